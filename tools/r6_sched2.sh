#!/bin/bash
# round 6: the scalar branch's ordering edges at the other shapes
{
python tools/ab_opt.py 1 2000 200 50 5 -- default tail_after=0 node_after=1 tail_after=0,node_after=1 tail_after=0,node_after=2
python tools/ab_opt.py 8 1000 200 50 5 -- default tail_after=0 tail_after=0,node_after=1
python tools/ab_opt.py 1 300 500 300 10 -- default tail_after=0 tail_after=0,node_after=2 tail_after=0,node_after=4
python tools/ab_opt.py 8 100 500 300 10 -- default tail_after=0 tail_after=0,node_after=2
python tools/ab_opt.py 1 100 2000 200 7 -- default tail_after=0 tail_after=0,node_after=4 tail_after=0,node_after=8
python tools/ab_opt.py 1 2000 70 19 5 -- default tail_after=0 tail_after=0,node_after=0
} > gpurun_out/r6_sched2.log 2>&1
cat gpurun_out/r6_sched2.log
