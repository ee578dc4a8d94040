#!/bin/bash
# copy the judged summaries of a final-measurement run (tools/measure_round.sh R) from gpurun_out/R into profiles/
R=$1; O=gpurun_out/$R
cp $O/bench.json profiles/${R}_bench.json
cp $O/bench_dp_selftest.json profiles/${R}_bench_dp_selftest.json
cp $O/train_step_kernel_stats.txt profiles/${R}_train_step_kernel_stats.txt
cp $O/train_step_single_stream_kernel_stats.txt profiles/${R}_train_step_kernel_stats_single_stream.txt
cp $O/encoder_decoder_only_kernel_stats.txt profiles/${R}_encoder_decoder_only_kernel_stats.txt
cp $O/encoder_decoder_only_single_stream_kernel_stats.txt profiles/${R}_encoder_decoder_only_kernel_stats_single_stream.txt
cp $O/train_step_mfma_pmc.txt profiles/${R}_train_step_mfma_pmc.txt
cp $O/encoder_decoder_only_mfma_pmc.txt profiles/${R}_encoder_decoder_only_mfma_pmc.txt
cp $O/${R}_mfma_busy.json profiles/${R}_mfma_busy.json
cp $O/decode_kernel_stats.txt profiles/${R}_decode_kernel_stats.txt
[ -f $O/decode_fp32_kernel_stats.txt ] && cp $O/decode_fp32_kernel_stats.txt profiles/${R}_decode_fp32_kernel_stats.txt
cp $O/knn_kernel_stats.txt profiles/${R}_knn_kernel_stats.txt
cp $O/knn_microbench.txt profiles/${R}_knn_microbench.txt
cp $O/${R}_hbm_traffic.json profiles/${R}_hbm_traffic.json
cp $O/train_step_hbm_traffic_pmc.txt profiles/${R}_train_step_hbm_traffic_pmc.txt
cp $O/knn_pmc.txt profiles/${R}_knn_pmc.txt
for f in loop_phases_lag1 loop_phases_lag0 at_bench knn_two_stage_stages knn_filtered_ab h2d_probe sample_whole; do [ -f $O/$f.txt ] && grep -v "amdgpu.ids" $O/$f.txt > profiles/${R}_$f.txt; done
ls -la profiles | grep $R
