"""gnntf on MI355X: the reference's flat namespace (reference gnntf/__init__.py:1-2) over a
HIP propagation path.  ``import gnntf`` then ``gnntf.APPNP``, ``gnntf.graph2adj``,
``gnntf.NodeClassification`` ... work as in the reference's README and demos."""
from .metrics import set_seed, acc, auc, avprec, rec, prec, f1
from .params import WrappedVariable, VariableGenerator, set_default_device, default_device
from .protocol import Layer, Layered
from .blocks import Dense, Dropout, Activation, Branch, Resume, Concatenate, Tradeoff, relu, linear
from .training import Predictor, Trainable
from .tasks import NodeClassification
from .link_tasks import LinkPrediction, MeanLinkPrediction, negative_sampling, recommend_all
from .sparse import (SparseCOO, DeviceGraph, Adjacency, SparseRows, spmm, spmm_bias_act, ppr_step, ppr_loop, appnp_propagate, gather_rows, normalize,
                     as_coo, dense, sparse_dense, gcnii_step, node_ce, node_argmax, edge_scores)
from .graph_io import create_nx_graph, adj2graph, graph2indices, graph2adj
from .graph_model import (MLP, GNN, Structural, NGCFLayer, NGCF, PPRIteration, PPRLoop, APPNP, GCNLayer, GCNSpectralPreservingLayer, GCN, GCNIILayer,
                          GCNIISpectralPreservingLayer, GCNII)
from .datasets import load_npz, save_npz

__version__ = "0.1.0"
