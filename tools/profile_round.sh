#!/bin/bash
# Regenerates the profile artefacts of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r02_v1      -> gpurun_out/prof_r02_v1/*  (copy the summaries into profiles/rNN/ afterwards)
# rocprofv3 gets the python interpreter directly after `--` (no wrapper hop); PMC counters in separate passes.
set -eo pipefail
tag=${1:-vXX}
out=gpurun_out/prof_$tag
mkdir -p $out
# whatever happens, leave only small files behind (gpurun copies back at most 64 MiB)
trap 'find $out -name "*.db" -delete; find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete; find $out -name "*agent_info.csv" -delete' EXIT
export TMPDIR=/tmp
# PMC passes FIRST (separate runs, counters only), and their stamped summaries into profiles/ of this tree, so that the bench
# lines below report the traffic / matrix-pipe figures of the code they time (bench.py prints traffic_stale otherwise)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python tools/launch_table.py --reps 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python tools/launch_table.py --reps 1 > /dev/null 2>&1
python tools/pmc_traffic.py "$(find $out/fetch -name '*counter_collection.csv' | head -1)" \
       "$(find $out/write -name '*counter_collection.csv' | head -1)" -o $out/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $out/sq -o sq -- python tools/launch_table.py --reps 1 > /dev/null 2>&1
python tools/pmc_mfma.py "$(find $out/sq -name '*counter_collection.csv' | head -1)" -o $out/pmc_mfma_busy.json > $out/pmc_mfma_busy.txt
cp $out/pmc_traffic.json $out/pmc_mfma_busy.json profiles/
find $out -name "*counter_collection.csv" -delete
python bench.py > $out/${tag}_default_bench.json
python bench.py --config 5 --no-configs > $out/${tag}_config5_bench.json
python bench.py --config 5 --config5-inflight 1 --no-cpu-baseline --no-configs > $out/${tag}_config5_one_in_flight_bench.json
python bench.py --batch 1 --no-cpu-baseline --no-variants --no-configs > $out/${tag}_config2_batch1_bench.json
python tools/launch_table.py > $out/${tag}_launch_table.txt 2>&1
python tools/wgtrace.py > $out/${tag}_wgtrace_inflight4.txt 2>&1
python tools/wgtrace.py --ablate knn > $out/${tag}_wgtrace_noknn.txt 2>&1
python tools/wgtrace.py --ablate fps,knn > $out/${tag}_wgtrace_mlp_only.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/default -o default -- python bench.py --no-cpu-baseline --no-variants --no-configs --repeats 3 > $out/${tag}_default_bench_under_rocprof.json 2> $out/rocprof_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/eager -o eager -- python bench.py --launch eager --inflight 1 --no-cpu-baseline --no-variants --no-configs --repeats 3 > $out/${tag}_eager_inflight1_under_rocprof.json 2> $out/rocprof_eager.err
# config 5 under the profiler, COOPERATIVE launch of the large-cloud sampler (one batch in flight).  rocprofv3 writes its
# outputs and finalises, THEN the process faults inside its exit handlers (profiles/r03/README.md: the only code of
# libpwclo_hip.so that runs at exit() is the compiler-generated __hip_module_dtor -> __hipUnregisterFatBinary, after the
# tool has already torn the runtime down): the exit status is ignored, the kernel statistics are what is kept.
rocprofv3 --kernel-trace --stats --output-format csv -d $out/c5 -o c5 -- python bench.py --config 5 --config5-inflight 1 --no-cpu-baseline --no-configs > $out/${tag}_config5_coop_under_rocprof.json 2> $out/rocprof_c5.err || echo "config-5 rocprof pass: exit status $? (see rocprof_c5.err; the statistics file is checked below)"
find $out -name "*kernel_stats.csv" | sort
cp "$(find $out/default -name '*kernel_stats.csv' | head -1)" $out/${tag}_default_bench_kernel_stats.csv
cp "$(find $out/eager -name '*kernel_stats.csv' | head -1)" $out/${tag}_eager_inflight1_kernel_stats.csv
cp "$(find $out/c5 -name '*kernel_stats.csv' | head -1)" $out/${tag}_config5_kernel_stats.csv || true
# training step of configs[3]'s per-GPU share: eager and graphed throughput, per-layer convolution table,
# rocprofv3 kernel summary
python tools/train_step.py --batch 32 --steps 10 --warmup 3 --fused-adam > $out/${tag}_train_step_b32.jsonl 2> /dev/null
python tools/train_step.py --batch 32 --steps 10 --warmup 3 --fused-adam --graph >> $out/${tag}_train_step_b32.jsonl 2> /dev/null
python tools/conv_table.py > $out/${tag}_train_conv_table.txt 2>&1
bash tools/profile_train.sh 32 > /dev/null 2>&1 && cp gpurun_out/prof_train/summary.txt $out/${tag}_train_step_b32_kernel_summary.txt
bash tools/gp_profile.sh $out/${tag}_group_points_roofline.txt > /dev/null 2>&1 || echo "group_points roofline pass failed"
python tools/train_parity_diag.py > $out/${tag}_train_parity_diag.txt 2>&1 || echo "train parity diagnostic failed"
tail -c 300 $out/${tag}_default_bench_under_rocprof.json; echo
ls -la $out
