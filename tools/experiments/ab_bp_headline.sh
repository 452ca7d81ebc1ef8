#!/bin/bash
# headline shape (n=500, V=100, R=7), 8 chains and one chain: the back-projection kernels interleaved (us per sweep)
python tools/ab_opt.py 8 400 500 100 7 -- cu_backproj=0 cu_backproj=1
python tools/ab_opt.py 1 400 500 300 10 -- cu_backproj=0,pair_backproj=0 cu_backproj=1
python tools/ab_opt.py 1 300 2000 200 7 -- cu_backproj=0 cu_backproj=1
