#!/bin/bash
# whole-step time against the tuning knobs that are read from the environment
O=gpurun_out/knobs; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 3 --skip-cpu --skip-knn --skip-decode > $O/$tag.json 2> $O/$tag.err; python -c "
import json; d=json.load(open('$O/$tag.json')); print('$tag', round(d['ms_per_step'],3), round(d['roofline_split']['encoder_decoder']['ms'],3))"; }
run base A=1
run wg64 RALF_WGRAD_GROUP_TILES=64
run wg96 RALF_WGRAD_GROUP_TILES=96
run wg128 RALF_WGRAD_GROUP_TILES=128
run wg96s2048 RALF_WGRAD_GROUP_TILES=96 RALF_WGRAD_GROUP_WGS=2048
run wg96s1536 RALF_WGRAD_GROUP_TILES=96 RALF_WGRAD_GROUP_WGS=1536
run wg128s2048 RALF_WGRAD_GROUP_TILES=128 RALF_WGRAD_GROUP_WGS=2048
run base2 A=1
run wg96b RALF_WGRAD_GROUP_TILES=96
