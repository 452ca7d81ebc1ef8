#!/bin/bash
# round 6: where to order the scalar branch behind the critical chain (options tail_after / node_after), 8 chains and one chain at the headline shape
{
python tools/ab_opt.py 8 640 500 100 7 -- default tail_after=0 node_after=0 node_after=1 node_after=2 node_after=3 node_after=4 node_after=6 tail_after=0,node_after=1 tail_after=0,node_after=2 tail_after=0,node_after=3 tail_after=0,node_after=4
python tools/ab_opt.py 1 1000 500 100 7 -- default tail_after=0 node_after=0 node_after=1 node_after=2 node_after=3 node_after=4 node_after=6 tail_after=0,node_after=2 tail_after=0,node_after=4
} > gpurun_out/r6_sched.log 2>&1
cat gpurun_out/r6_sched.log
