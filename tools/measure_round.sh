#!/bin/bash
# final measurements of the round: bench line, rocprofv3 kernel statistics, PMC HBM traffic, kNN / decode profiles
R=${1:-r04a}
O=gpurun_out/$R; mkdir -p $O
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
prof() { # name, steps-in-trace, header, command...
  name=$1; steps=$2; header=$3; shift 3
  rm -rf $ROOT/$O/p_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$O/p_$name -o t -- "$@" > $ROOT/$O/${name}.log 2>&1
  DB=$(find $ROOT/$O/p_$name -name "*.db" | head -1)
  python3 $ROOT/tools/prof_summary.py $DB $ROOT/$O/${name}_kernel_stats.txt "$header" $steps $GAP
  rm -rf $ROOT/$O/p_$name
}
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants --profile-pause 1"
# (the window after the idle second = the 5 timed graph replays only: no model set-up, eager warm-up, snapshot copies or capture)
GAP=300 prof train_step 5 "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants --profile-pause 1 (MI355X, $R; graph replay; the parameter-gradient kernels are a second graph on the side stream: per-kernel times include overlap, and the profiler's per-launch host cost lets that graph start late, so the wall time per step in THIS trace is 3-4 ms above the unprofiled step of bench.json)" $BENCH
GAP=300 prof train_step_single_stream 5 "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 ... --profile-pause 1 --no-overlap (MI355X, $R; one stream: per-kernel times are not inflated by overlap)" $BENCH --no-overlap
prof encoder_decoder_only 11 "# rocprofv3 --kernel-trace --stats -- python3 tools/encdec_once.py 8 (MI355X, $R; backbone replaced by a fixed feature sequence; whole trace incl. eager warm-up + capture)" python3 $ROOT/tools/encdec_once.py 8
# single stream, graph replays only (the window after the idle second): kernel time per step re-derives the HIP-event figure of bench.py
GAP=300 prof encoder_decoder_only_single_stream 8 "# rocprofv3 --kernel-trace --stats -- python3 tools/encdec_once.py 8 single (MI355X, $R; backbone replaced by a fixed feature sequence; ONE stream: no parallel graph branches, kernel durations not inflated by overlap)" python3 $ROOT/tools/encdec_once.py 8 single
prof decode 5 "# rocprofv3 --kernel-trace --stats -- python3 tools/decode_once.py 3 (MI355X, $R; B = 256, task c, argmax; 4 replays of the captured loop + 2 eager warm-up passes in the trace)" python3 $ROOT/tools/decode_once.py 3
prof decode_fp32 5 "# rocprofv3 --kernel-trace --stats -- python3 tools/decode_once.py 3 float32 (MI355X, $R; B = 256, task c, argmax, the fp32 parity mode; 4 replays of the captured loop + 2 eager warm-up passes in the trace)" python3 $ROOT/tools/decode_once.py 3 float32
prof knn 1 "# rocprofv3 --kernel-trace --stats -- python3 tools/knn_once.py (MI355X, $R; 61548 x 1792 fp32, nq = 16: 20 scans + 10 whole calls, nq = 1024: 3 scans + 1 whole call)" python3 $ROOT/tools/knn_once.py
# PMC passes (one counter per run; --kernel-trace only)
PB="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-graph --skip-cpu --skip-knn --skip-split --skip-decode --skip-variants"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/$O/pmc_$C $ROOT/$O/pmck_$C
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d $ROOT/$O/pmc_$C -o t -- $PB > $ROOT/$O/pmc_$C.log 2>&1
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $ROOT/$O/pmck_$C -o t -- python3 $ROOT/tools/knn_once.py > $ROOT/$O/pmck_$C.log 2>&1
done
# MFMA utilisation (one pass, SQ + GRBM counters fit together): whole train step and encoder-decoder-only step, eager
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
rm -rf $ROOT/$O/pmc_mfma $ROOT/$O/pmc_mfma_ed
timeout 900 rocprofv3 --pmc $MF --kernel-trace -d $ROOT/$O/pmc_mfma -o t -- $PB > $ROOT/$O/pmc_mfma.log 2>&1
timeout 900 rocprofv3 --pmc $MF --kernel-trace -d $ROOT/$O/pmc_mfma_ed -o t -- python3 $ROOT/tools/encdec_once.py 2 single eager > $ROOT/$O/pmc_mfma_ed.log 2>&1
cd $ROOT
DB=$(find $O/pmc_mfma -name "*.db" | head -1)
python3 tools/pmc_mfma.py $DB $O/train_step_mfma_pmc.txt "# rocprofv3 --pmc $MF --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-graph --skip-cpu --skip-knn --skip-split --skip-decode (MI355X, $R; eager so that every dispatch is attributed; 3 train steps in the trace, B=64, bf16, N=10)" 3 $O/${R}_mfma_busy.json train_step_B64_N10_bf16
DB=$(find $O/pmc_mfma_ed -name "*.db" | head -1)
python3 tools/pmc_mfma.py $DB $O/encoder_decoder_only_mfma_pmc.txt "# rocprofv3 --pmc $MF --kernel-trace -- python3 tools/encdec_once.py 2 single eager (MI355X, $R; backbone replaced by a fixed feature sequence; 3 eager steps in the trace)" 3 $O/${R}_mfma_busy.json encoder_decoder_B64_N10_bf16
rm -rf $O/pmc_mfma $O/pmc_mfma_ed
cp $O/${R}_mfma_busy.json profiles/${R}_mfma_busy.json
F=$(find $O/pmc_FETCH_SIZE -name "*.db" | head -1); W=$(find $O/pmc_WRITE_SIZE -name "*.db" | head -1)
python3 tools/pmc_traffic_json.py $O/${R}_hbm_traffic.json train_step_B64_N10_bf16 $F $W 3 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-graph --skip-cpu --skip-knn --skip-split --skip-decode (separate passes; 3 eager steps in each trace, model set-up included)" > $O/pmc_summary.txt 2>&1
python3 tools/pmc_total.py $F FETCH_SIZE 3 > $O/train_step_hbm_traffic_pmc.txt 2>&1; python3 tools/pmc_total.py $W WRITE_SIZE 3 >> $O/train_step_hbm_traffic_pmc.txt 2>&1
F=$(find $O/pmck_FETCH_SIZE -name "*.db" | head -1); W=$(find $O/pmck_WRITE_SIZE -name "*.db" | head -1)
python3 tools/pmc_traffic_json.py $O/${R}_hbm_traffic.json knn_scores_nq16_61548x1792 $F $W 1 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/knn_once.py (separate passes; per dispatch of knn_scores_kernel<16,1,2>)" "knn_scores_kernel%16%1%2" >> $O/pmc_summary.txt 2>&1
python3 tools/pmc_traffic_json.py $O/${R}_hbm_traffic.json knn_scores_nq1024_61548x1792 $F $W 1 "same passes; per dispatch of knn_scores_kernel<32,4,2>" "knn_scores_kernel%32%4%2" >> $O/pmc_summary.txt 2>&1
python3 tools/pmc_total.py $F FETCH_SIZE 1 > $O/knn_pmc.txt 2>&1; python3 tools/pmc_total.py $W WRITE_SIZE 1 >> $O/knn_pmc.txt 2>&1
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmck_FETCH_SIZE $O/pmck_WRITE_SIZE
cat $O/pmc_summary.txt
# the new traffic file must be visible to bench.py in THIS run
cp $O/${R}_hbm_traffic.json profiles/${R}_hbm_traffic.json
timeout 300 python3 tools/knn_graph_bench.py 1 16 32 64 128 1024 > $O/knn_microbench.txt 2>&1
timeout 2400 python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
timeout 600 python3 bench.py --dp-selftest --skip-cpu --skip-knn --skip-split --skip-decode > $O/bench_dp_selftest.json 2> $O/bench_dp.err; python3 -c "
import json; d=json.load(open('$O/bench_dp_selftest.json')); print('dp-selftest ms', d['ms_per_step'])"

# round 4: the unchanged reference loop (phases), the BatchNorm-fusion microbenchmark, the two-stage kNN stages, the host link
timeout 200 python3 tools/loop_phases.py 1 > $O/loop_phases_lag1.txt 2>&1; timeout 200 python3 tools/loop_phases.py 0 > $O/loop_phases_lag0.txt 2>&1
timeout 300 python3 tools/at_bench.py > $O/at_bench.txt 2>&1
timeout 300 python3 tools/knn_two_stage_probe.py > $O/knn_two_stage_stages.txt 2>&1
timeout 300 python3 tools/knn_filtered_ab.py 512 640 1024 2048 2>&1 | grep 'nq =' > $O/knn_filtered_ab.txt
timeout 100 python3 tools/h2d_probe.py > $O/h2d_probe.txt 2>&1
timeout 300 python3 tools/sample_phases.py 15 whole 2>&1 | grep -E 'GATES|picked|whole' > $O/sample_whole.txt; RALF_DECODE_GATES=0 timeout 300 python3 tools/sample_phases.py 15 whole 2>&1 | grep -E 'GATES|whole' >> $O/sample_whole.txt
