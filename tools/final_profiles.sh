#!/bin/bash
# Round-end measurement pass on the GPU box: GPU tests, race screens, the bench line, rocprofv3 kernel stats (two-stream
# and single-stream), the two PMC passes behind roofline.traffic, a kernel-trace timeline of steady-state steps, and the
# other configurations (config 4, config 5, hipGraph replay). Results under gpurun_out/$1 (default: final).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-final}
PART=${2:-all}          # a | b | all: the pass in two gpurun calls (each under the 20-minute limit of a call)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$PART" != "b" ]; then
echo "== pytest"; timeout -k 10 900 python3 -m pytest "$R/tests" -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; tail -n 2 "$OUT/pytest_gpu.log"
echo "== race screens"
RUNS=40 STEPS=24 timeout -k 10 200 python3 "$R/tools/debug_race_steps.py" 2>/dev/null | tee "$OUT/race_steps.log" || exit 1
RUNS=600 timeout -k 10 300 python3 "$R/tools/debug_race.py" 2>/dev/null | tail -n 1 | tee "$OUT/race_step.log" || exit 1
echo "== bench"; timeout -k 10 400 python3 "$R/bench.py" > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cut -c1-330 "$OUT/bench.json"
echo "== phase times"; timeout -k 10 100 python3 "$R/tools/phase_times.py" 2>/dev/null | tee "$OUT/phase_times.log"
echo "== rocprof stats (default command: timed region on two streams + the two event passes)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p1" -- python3 "$R/bench.py" --steps 10 --warmup 3 --settle 0 --no-cpu-baseline --no-other-configs > "$OUT/p1.log" 2>&1 || exit 1
echo "== rocprof stats (timed region only, two streams)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p3" -- python3 "$R/bench.py" --steps 10 --warmup 3 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/p3.log" 2>&1 || exit 1
echo "== rocprof stats (single stream)"
INSAR_SIDE_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/p2" -- python3 "$R/bench.py" --allow-switches --steps 10 --warmup 3 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/p2.log" 2>&1 || exit 1
echo "== PMC passes (launch configuration of the timed region: split-K factors for the side stream; the profiler serialises the kernels)"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pF" -- python3 "$R/bench.py" --steps 4 --warmup 1 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/pF.log" 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pW" -- python3 "$R/bench.py" --steps 4 --warmup 1 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/pW.log" 2>&1 || exit 1
echo "== kernel trace (timeline of 4 steady-state steps)"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pT" -- python3 "$R/bench.py" --steps 4 --warmup 3 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/pT.log" 2>&1 || exit 1
cp $(ls "$OUT"/p1/*/*_kernel_stats.csv | head -n 1) "$OUT/bench_kernel_stats.csv"
cp $(ls "$OUT"/p3/*/*_kernel_stats.csv | head -n 1) "$OUT/bench_kernel_stats_timed_region.csv"
cp $(ls "$OUT"/p2/*/*_kernel_stats.csv | head -n 1) "$OUT/bench_kernel_stats_single_stream.csv"
cp $(ls "$OUT"/pF/*/*_counter_collection.csv | head -n 1) "$OUT/pmc_fetch.csv"
cp $(ls "$OUT"/pW/*/*_counter_collection.csv | head -n 1) "$OUT/pmc_write.csv"
cp $(ls "$OUT"/pT/*/*_kernel_trace.csv | head -n 1) "$OUT/kernel_trace.csv"
python3 "$R/tools/pmc_traffic.py" "$OUT/pmc_fetch.csv" "$OUT/pmc_write.csv" "$OUT/pmc_traffic.json" "timed region (weight gradients split for engine.WGRAD_FILL = 0.5 of the work-group slots, as beside the dgrad chain)" | tee "$OUT/pmc_traffic.txt"
python3 "$R/tools/timeline.py" "$OUT/kernel_trace.csv" | tee "$OUT/timeline.txt"
rm -rf "$OUT/p1" "$OUT/p2" "$OUT/p3" "$OUT/pF" "$OUT/pW" "$OUT/pT" "$OUT/pmc_fetch.csv" "$OUT/pmc_write.csv"      # keep the summaries only
fi
if [ "$PART" = "a" ]; then echo "part a done"; exit 0; fi
echo "== SQ counters (one PMC pass, single stream: every kernel alone on the chip)"
INSAR_SIDE_STREAM=0 timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pS" -- python3 "$R/bench.py" --allow-switches --steps 3 --warmup 1 --settle 0 --no-cpu-baseline --no-kernel-timing --no-other-configs > "$OUT/pS.log" 2>&1 && python3 "$R/tools/sq_counters.py" $(ls "$OUT"/pS/*/*_counter_collection.csv | head -n 1) 1e8 | tee "$OUT/pmc_sq_counters.txt"
rm -rf "$OUT/pS"
echo "== in-kernel stamps (diagnostic build of the library)"
if [ -f "$R/insar_unet_ca_amd/libinsar_hip_stamps.so" ]; then
  INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_stamps.so timeout -k 10 200 python3 "$R/tools/stamp_gemm.py" 2>/dev/null | tee "$OUT/stamps_gemm.txt"
  INSAR_HIP_LIB=$R/insar_unet_ca_amd/libinsar_hip_stamps.so timeout -k 10 200 python3 "$R/tools/stamp_flat.py" 2>/dev/null | tee "$OUT/stamps_flat.txt"
fi
echo "== stream-input vs resident"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["value"])'
for i in 1 2; do
  timeout -k 10 150 python3 "$R/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs 2>/dev/null | python3 -c "$pick" resident | tee -a "$OUT/stream_input.txt"
  timeout -k 10 150 python3 "$R/bench.py" --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-other-configs --stream-input 2>/dev/null | python3 -c "$pick" streamed | tee -a "$OUT/stream_input.txt"
done
echo "== hipGraph replay vs eager launches"
timeout -k 10 300 python3 "$R/bench.py" --graph on --no-cpu-baseline --no-kernel-timing --no-other-configs --steps 30 --warmup 5 > "$OUT/bench_graph.json" 2> "$OUT/bench_graph.err"; cut -c1-200 "$OUT/bench_graph.json"
timeout -k 10 300 python3 "$R/bench.py" --graph off --no-cpu-baseline --no-kernel-timing --no-other-configs --steps 30 --warmup 5 > "$OUT/bench_eager.json" 2> "$OUT/bench_eager.err"; cut -c1-200 "$OUT/bench_eager.json"
echo "== microbench: ping-pong K loop, same-process A/B"
timeout -k 10 300 python3 "$R/tools/gemm_bench.py" --only down2.3,conv2.0,conv1.0 --what pp 2>/dev/null | tee "$OUT/gemm_pingpong_ab.txt"
echo "== microbench: 256 x 128 six-phase weight gradient (wgrad3x.hip) against the 128-tile kernel, every layer it serves"
timeout -k 10 300 python3 "$R/tools/gemm_bench.py" --what x3 --only down2.3,conv2.0,down3.0,down3.3,conv1.0,down4.0,down4.3,conv3.0,down2.0 2>/dev/null | tee "$OUT/wgrad3x_microbench.txt"
echo "== 2-rank rehearsal (gloo, both ranks on the one GPU: exercises the bucketed reducer inside backward)"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 "$R/bench.py" --gpus 2 --steps 3 --warmup 1 --backend gloo --no-cpu-baseline --no-kernel-timing > "$OUT/bench_dp2_gloo.json" 2> "$OUT/bench_dp2_gloo.err"; cut -c1-160 "$OUT/bench_dp2_gloo.json"
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 "$R/bench.py" --gpus 2 --steps 3 --warmup 1 --backend gloo --shard-optimizer --no-cpu-baseline --no-kernel-timing > "$OUT/bench_dp2_gloo_sharded.json" 2> "$OUT/bench_dp2_gloo_sharded.err"; cut -c1-160 "$OUT/bench_dp2_gloo_sharded.json"
echo "== config 4"; timeout -k 10 300 python3 "$R/bench.py" --dtype f32 --size 512 --batch 8 --loss ce --no-cpu-baseline > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"; cut -c1-200 "$OUT/bench_cfg4.json"
echo "== config 5"; timeout -k 10 300 python3 "$R/bench.py" --model deeplab --no-cpu-baseline > "$OUT/bench_cfg5.json" 2> "$OUT/bench_cfg5.err"; cut -c1-200 "$OUT/bench_cfg5.json"
echo done
echo "== stand-alone pass timings"; timeout -k 10 300 python3 "$R/tools/pass_bench.py" 2>/dev/null | tee "$OUT/pass_bench.txt"
