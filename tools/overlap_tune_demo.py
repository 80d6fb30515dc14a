#!/usr/bin/env python3
"""Comm.tune_overlap right after the process group is made (one-rank "nccl" group as its own peer): candidate exchange-lane
streams (and, when none hides, further process groups), each timed with a 1 GiB RCCL send / recv to itself beside matrix products
of the compute stream.  About one stream in four shares the compute stream's hardware queue and hides nothing.  Prints the table.
ORDER=streams_first creates a few streams before the process group (another placement of torch's communicator stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gnn-tf_amd")]
import torch, torch.distributed as dist
fd = os.dup(1); os.dup2(2, 1)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
_pre = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("PRE_STREAMS", "0")))]
dist.init_process_group("nccl", device_id=dev)
from gnntf import sharded
import json
table = sharded.Comm(group=None).tune_overlap(dev, force=True)
dist.destroy_process_group()
os.write(fd, (json.dumps(table, indent=1) + "\n").encode())
