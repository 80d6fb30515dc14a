#!/bin/bash
# Round-4 GPU visit 4: PMC passes of rank 0's vertex block at P = 2 / 4 / 8 (the N > 1 line's roofline.traffic), then reduced-size
# rehearsals of the all-auto bench command on gloo ranks sharing the card.
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
for cfg in "8 cover 2" "8 cover 4" "8 pull 2" "4 cover 2" "2 cover 2"; do
  set -- $cfg; P=$1; COVER=$2; CH=$3
  T=p${P}_${COVER}_c${CH}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/${T}_$ctr -o run -- python3 tools/sim_blocks.py --world $P --cover $COVER --chunks $CH --pmc-iterations 5 \
        > $O/${T}_$ctr.json 2> $O/${T}_$ctr.err || { echo "$T $ctr failed"; tail -5 $O/${T}_$ctr.err; exit 1; }
  done
  echo "$T done: $(tail -c 300 $O/${T}_FETCH_SIZE.json)"
done
bash tools/rehearse_bench.sh $O 2 5 || exit 1
echo "all done"
