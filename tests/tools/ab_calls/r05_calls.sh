#!/bin/bash
# The GPU leases of round 5 as they were run: `gpurun -- bash tests/tools/ab_calls/r05_calls.sh <n>`.  Each case is one lease (smoke guard,
# parity, in-call A/B through tests/tools/ab_run.sh, stamps / per-kernel profiles); results: profiles/r05_ab_calls.md.  Variant libraries
# (video_prediction_amd/ab/libsavp_hip_<tag>.so) come from tests/tools/build_variant.sh and are not kept in the tree.
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
t0=$(date +%s)
smoke() { python bench.py --steps 2 --warmup 2 --no-f32 --no-cpu-baseline --inst-steps 1 > $O/smoke.json 2> $O/smoke.err || { echo "SMOKE FAILED"; tail -25 $O/smoke.err; exit 1; }; }
case "$1" in
1)
  # Round 5, first lease: (i) the new replayed-vs-eager test at the bench shapes (records the distances of the CURRENT kernels: float
  # atomics), (ii) parity of the two patches staged by round 4 (ring lane table, generic-conv fragment pre-read) on every shipped tuning-table
  # instantiation, (iii) their in-call A/B.
  O=gpurun_out/r05a; mkdir -p $O
  smoke
  timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replayed_bench_step" > $O/replay_test.log 2>&1; echo "replay test rc=$? $(( $(date +%s)-t0 ))s"; tail -15 $O/replay_test.log | cut -c1-600
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_both.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "table or conv" > $O/both_ops.log 2>&1; echo "both ops rc=$? $(( $(date +%s)-t0 ))s"; tail -4 $O/both_ops.log | cut -c1-400
  OUT=$O REPS=2 bash tests/tools/ab_run.sh base "" lanetab "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_lanetab.so" preread "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_preread.so" both "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_both.so"
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
2)
  # Round 5, second lease: the deterministic-reduction build -- whole GPU suite (incl. the replayed-vs-eager test at the bench shapes), the
  # round-4 library against it in one call, and cycle stamps of the shipped gate-convolution instantiations (stamps-only developer build).
  O=gpurun_out/r05b; mkdir -p $O
  smoke
  timeout 1200 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_model.py::test_the_replayed_bench_step_is_the_eager_step_and_matches_the_golden > $O/gputest.log 2>&1; echo "gputest rc=$? $(( $(date +%s)-t0 ))s"; tail -12 $O/gputest.log | cut -c1-400
  timeout 600 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replayed_bench_step" > $O/replay_test.log 2>&1; echo "replay test rc=$? $(( $(date +%s)-t0 ))s"; grep -E "^E  |passed|failed" $O/replay_test.log | cut -c1-300 | head -20
  for c in c2 c4 c5; do cat gpurun_out/r05_replay_vs_eager_$c.json | tr -d '\n'; echo; done
  OUT=$O REPS=2 bash tests/tools/ab_run.sh r04 "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_r04.so" det ""
  for spec in lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:711:src16 lstm_h2:dgrad:311:src16; do
    for blk in 0 100 200; do
      RING_BLOCK=$blk SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v amdgpu.ids | sed "s/^/blk$blk /"
    done
  done | tee $O/ring_stamps.log
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
3)
  # Round 5, third lease: row-wise LDS-DMA patch staging in the ring kernel -- parity (every shipped tuning-table instantiation, the bench-shape
  # goldens now under ONE gradient gate, the replayed-vs-eager tests with their measured gates), in-call A/B against the build before it, stamps.
  O=gpurun_out/r05c; mkdir -p $O
  smoke
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -5 $O/ops.log | cut -c1-300
  timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replay or golden or hipgraph or recipe_shapes" > $O/model.log 2>&1; echo "model rc=$? $(( $(date +%s)-t0 ))s"; grep -E "^E  .*(Assert|assert)|passed|failed" $O/model.log | cut -c1-600 | head -20
  for c in c2 c4 c5; do cat gpurun_out/r05_replay_vs_eager_$c.json | tr -d '\n'; echo; done
  OUT=$O REPS=2 bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma ""
  OUT=$O/c5 REPS=1 CONFIG=c5 BENCH_ARGS="--steps 20 --warmup 4 --no-f32 --no-cpu-baseline --inst-steps 4" bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma ""
  for spec in lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:711:src16; do
    SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v amdgpu.ids
  done | tee $O/ring_stamps.log
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
4)
  # Round 5, fourth lease: per-kernel view (rocprofv3 kernel stats of 6 eager steps) of the ring kernel's patch-staging variants -- slot-linear
  # (before), row-wise (default now), row-wise + first group requested before the rest of the prologue -- and the repaired golden / replay tests.
  O=gpurun_out/r05d; mkdir -p $O
  smoke
  timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -k "replayed_bench or golden" > $O/model.log 2>&1; echo "model rc=$? $(( $(date +%s)-t0 ))s"; grep -E "^E  .*(Assert|assert)|passed|failed" $O/model.log | cut -c1-700 | head -20
  for v in det rowdma earlypatch; do
    lib=$PWD/video_prediction_amd/ab/libsavp_hip_$v.so; [ $v = rowdma ] && lib=$PWD/video_prediction_amd/libsavp_hip.so
    SAVP_LIB=$lib bash tests/tools/prof_step.sh r05d/$v 2>&1 | tail -2
  done
  python tests/tools/compare_stats.py $O/det_kernel_stats.csv $O/rowdma_kernel_stats.csv 6 | tee $O/cmp_det_rowdma.txt
  python tests/tools/compare_stats.py $O/rowdma_kernel_stats.csv $O/earlypatch_kernel_stats.csv 6 | tee $O/cmp_rowdma_early.txt
  OUT=$O REPS=2 bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma "" early "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_earlypatch.so"
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
5)
  # Round 5, fifth lease: row-wise patch staging, second version (incremental walk, no per-j array) -- conv parity, per-kernel view and step A/B
  # against the slot-linear build.
  O=gpurun_out/r05f; mkdir -p $O
  smoke
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv or table or cell" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/ops.log | cut -c1-300
  for v in det rowdma3; do
    lib=$PWD/video_prediction_amd/ab/libsavp_hip_$v.so; [ $v = rowdma3 ] && lib=$PWD/video_prediction_amd/libsavp_hip.so
    SAVP_LIB=$lib bash tests/tools/prof_step.sh r05f/$v 2>&1 | tail -1
  done
  python tests/tools/compare_stats.py $O/det_kernel_stats.csv $O/rowdma3_kernel_stats.csv 6 | tee $O/cmp_det_rowdma3.txt
  OUT=$O REPS=2 bash tests/tools/ab_run.sh det "SAVP_LIB=video_prediction_amd/ab/libsavp_hip_det.so" rowdma3 ""
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
7)
  # Round 5, seventh lease: what the slab DMAs cost the ring kernel's main loop -- stamps-only builds with all / one / none of the LW slab DMA
  # instructions per wave and entry (timing builds, results wrong by construction).
  O=gpurun_out/r05g; mkdir -p $O
  for v in stamps stampsd1 stampsd0; do
    for spec in lstm_h0:fprop:712:cell16 lstm_h1:fprop:711:cell16 lstm_h2:fprop:311:cell16 lstm_h0:dgrad:711:src16; do
      SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_$v.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave" | sed "s/^/$v /"
    done
  done | tee $O/ring_dma_ablate.log
  ;;
8)
  # Round 5, eighth lease: wide weight slabs (8 / 9 k-steps per entry: the look-ahead of three entries then covers the LDS-DMA latency) against the
  # shipped instantiations of the 16x16 / 8x8 gate convolutions, stamps-only build, isolated launches.
  O=gpurun_out/r05h; mkdir -p $O
  for spec in lstm_h1:fprop:711:cell16 lstm_h1:fprop:1711:cell16 lstm_h1:fprop:1311:cell16 lstm_h1:fprop:312:cell16 lstm_h2:fprop:311:cell16 lstm_h2:fprop:1311:cell16 lstm_h2:fprop:1711:cell16 lstm_h1:dgrad:711:src16 lstm_h1:dgrad:1711:src16 lstm_h2:dgrad:311:src16 lstm_h2:dgrad:1311:src16; do
    SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave"
  done | tee $O/ring_wide.log
  ;;
9)
  # Round 5, ninth lease: eight-deep weight ring for the small-tile ring instantiations (option ring_deep) -- conv parity (every shipped table
  # instantiation runs with it), stamps of the 16x16 gate convolutions with / without, in-call A/B of the step, per-kernel view.
  O=gpurun_out/r05j; mkdir -p $O
  smoke
  timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "conv or table or cell" > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/ops.log | cut -c1-300
  for d in 0 1; do for spec in lstm_h1:fprop:711:cell16 lstm_h1:dgrad:711:src16 lstm_h0:dgrad:711:src16; do
    SAVP_RING_DEEP=$d SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_stamps.so python tests/tools/ring_times.py $spec 2>&1 | grep -v "amdgpu.ids\|per wave" | sed "s/^/deep$d /"
  done; done | tee $O/ring_deep_stamps.log
  for v in deep0 deep1; do
    SAVP_RING_DEEP=${v#deep} bash tests/tools/prof_step.sh r05j/$v 2>&1 | tail -1
  done
  python tests/tools/compare_stats.py $O/deep0_kernel_stats.csv $O/deep1_kernel_stats.csv 6 | tee $O/cmp_deep.txt
  OUT=$O REPS=2 bash tests/tools/ab_run.sh deep0 "SAVP_RING_DEEP=0" deep1 "SAVP_RING_DEEP=1"
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
12)
  # twelfth lease: in-step re-tune of the most expensive conv problems on this round's kernels (row-wise staging changed the NKS >= 3
  # instantiations only), c2 / c4 / c5, then the tuned table against the shipped one
  O=gpurun_out/r05l; mkdir -p $O
  smoke
  T=video_prediction_amd/tuning_gfx950_bf16.json
  cp $T $O/table_before.json
  python tests/tools/insitu_tune.py $O/table_c2.json 40 5 > $O/insitu_c2.log 2>&1; tail -3 $O/insitu_c2.log
  [ -s $O/table_c2.json ] && cp $O/table_c2.json $T
  CONFIG=c4 python tests/tools/insitu_tune.py $O/table_c4.json 16 4 > $O/insitu_c4.log 2>&1; tail -3 $O/insitu_c4.log
  [ -s $O/table_c4.json ] && cp $O/table_c4.json $T
  CONFIG=c5 python tests/tools/insitu_tune.py $O/table_c5.json 16 4 > $O/insitu_c5.log 2>&1; tail -3 $O/insitu_c5.log
  [ -s $O/table_c5.json ] && cp $O/table_c5.json $T
  cp $T $O/tuning_gfx950_bf16.json
  OUT=$O REPS=2 bash tests/tools/ab_run.sh shipped "ARGS=--tuning-table,$O/table_before.json" tuned ""
  ;;
13)
  # thirteenth lease: the LDS-patch weight gradient at the step's operands (both tensors bf16, 928 images): per-phase cycle stamps of
  # workgroup 0 / wave 0 (stamps build of conv_wgrad_patch.hip) and the launch times of the shipped build
  O=gpurun_out/r05m; mkdir -p $O
  smoke
  SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_wgstamps.so python tests/tools/wgp_times.py 2>&1 | grep -v amdgpu.ids | tee $O/wgp_stamps.log
  BF16=1 python tests/tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | tee $O/wgrad_bench.log
  ;;
14)
  # fourteenth lease: LDS-DMA staging of the weight gradient's bf16 operands (option wgp_dma): parity, launch times with / without at the
  # step's operands, stamps, in-call A/B of the step
  O=gpurun_out/${OUTDIR:-r05n}; mkdir -p $O
  smoke
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_gradient or bf16_activation or table" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -5 $O/ops.log | cut -c1-300
  for d in 1 0; do SAVP_WGP_DMA=$d BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dma$d /"; done | tee $O/wgrad_bench.log
  for d in 1 0; do SAVP_WGP_DMA=$d SAVP_LIB=$PWD/video_prediction_amd/ab/libsavp_hip_wgstamps.so python tests/tools/wgp_times.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dma$d /"; done | tee $O/wgp_stamps.log
  OUT=$O REPS=2 bash tests/tools/ab_run.sh dma0 "SAVP_WGP_DMA=0" dma1 "SAVP_WGP_DMA=1"
  ;;
16)
  O=gpurun_out/r05p; mkdir -p $O
  for cfg in 0 48 84 44; do SAVP_WGP_CFG=$cfg BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg$cfg /"; done | tee $O/wgrad_cfg_sweep.log
  for sp in 256 512 768; do SAVP_WGP_SPLIT=$sp BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/split$sp /"; done | tee -a $O/wgrad_cfg_sweep.log
  for sp in 512 768 1024; do SAVP_WGP_CFG=48 SAVP_WGP_SPLIT=$sp BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg48_split$sp /"; done | tee -a $O/wgrad_cfg_sweep.log
  ;;
18)
  # eighteenth lease: row-group order [tap][channel group] of the LDS-patch weight gradient (bank conflicts of the transpose reads), and the
  # DMA pieces spread between the MFMAs (variant build -DSAVP_WGP_INTERLEAVE): parity, isolated launches, stamps, counters, step A/B
  O=gpurun_out/r05r; mkdir -p $O
  AB=$PWD/video_prediction_amd/ab
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_gradient or bf16_activation or table" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/ops.log | cut -c1-300
  SAVP_LIB=$AB/libsavp_hip_wgil.so timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_gradient or bf16_activation" > $O/ops_il.log 2>&1; echo "ops interleave rc=$?"; tail -3 $O/ops_il.log | cut -c1-300
  for v in old new il; do
    case $v in old) L=$AB/libsavp_hip_wgold.so;; new) L=;; il) L=$AB/libsavp_hip_wgil.so;; esac
    SAVP_LIB=$L BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  done | tee $O/wgrad_bench.log
  for v in new il; do
    case $v in new) L=$AB/libsavp_hip_wgstamps.so;; il) L=$AB/libsavp_hip_wgilstamps.so;; esac
    SAVP_LIB=$L python tests/tools/wgp_times.py lstm_h0 lstm_h1 lstm_h2 head3x3 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  done | tee $O/wgp_stamps.log
  bash tests/tools/pmc_wgrad.sh r05r lstm_h1 > /dev/null 2>&1; grep -A4 "pass 1" $O/wgrad_counters.log
  OUT=$O REPS=2 bash tests/tools/ab_run.sh old "SAVP_LIB=$AB/libsavp_hip_wgold.so" new "" il "SAVP_LIB=$AB/libsavp_hip_wgil.so"
  ;;
19)
  # nineteenth lease: weight gradient, DMA requests two tiles ahead into three buffers, issued by the two halves of the workgroup at opposite
  # ends of an iteration: parity, isolated launches, stamps, step A/B against the previous build
  O=gpurun_out/${OUTDIR:-r05s}; mkdir -p $O
  AB=$PWD/video_prediction_amd/ab
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_gradient or bf16_activation or table" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/ops.log | cut -c1-300
  for v in prev new; do
    case $v in prev) L=$AB/libsavp_hip_wgprev.so;; new) L=;; esac
    SAVP_LIB=$L BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  done | tee $O/wgrad_bench.log
  SAVP_LIB=$AB/libsavp_hip_wgstamps.so python tests/tools/wgp_times.py lstm_h0 lstm_h1 lstm_h2 head3x3 2>&1 | grep -v amdgpu.ids | tee $O/wgp_stamps.log
  OUT=$O REPS=2 bash tests/tools/ab_run.sh prev "SAVP_LIB=$AB/libsavp_hip_wgprev.so" new ""
  ;;
20)
  # twentieth lease: de-phased weight gradient with a deeper fragment read-ahead (6 / 4 MFMAs) against the lock-step build (prev)
  O=gpurun_out/${OUTDIR:-r05t}; mkdir -p $O
  AB=$PWD/video_prediction_amd/ab
  timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "weight_gradient or bf16_activation" > $O/ops.log 2>&1; echo "ops rc=$?"; tail -3 $O/ops.log | cut -c1-300
  for v in prev pd6 pd4; do
    case $v in prev) L=$AB/libsavp_hip_wgprev.so;; pd6) L=;; pd4) L=$AB/libsavp_hip_wgpd4.so;; esac
    SAVP_LIB=$L BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/$v /"
  done | tee $O/wgrad_bench.log
  SAVP_LIB=$AB/libsavp_hip_wgstamps.so python tests/tools/wgp_times.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | tee $O/wgp_stamps.log
  ;;
21)
  # lease 21: the tile descriptor table on the register-staged weight gradient too (fp32 operands): the whole op suite, isolated launches with
  # fp32 / bf16 operands, stamps, step A/B against the previous build (DMA + table for bf16 operands only)
  O=gpurun_out/${OUTDIR:-r05u}; mkdir -p $O
  AB=$PWD/video_prediction_amd/ab
  t0=$(date +%s)
  timeout 1500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x > $O/ops.log 2>&1; echo "ops rc=$? $(( $(date +%s)-t0 ))s"; tail -3 $O/ops.log | cut -c1-300
  for v in prev new; do
    case $v in prev) L=$AB/libsavp_hip_wgprev.so;; new) L=;; esac
    SAVP_LIB=$L BF16=0 python tests/tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v f32 /"
    SAVP_LIB=$L BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids | sed "s/^/$v bf16 /"
  done | tee $O/wgrad_bench.log
  BF16=0 SAVP_LIB=$AB/libsavp_hip_wgstamps.so python tests/tools/wgp_times.py 2>&1 | grep -v amdgpu.ids | sed "s/^/f32 /" | tee $O/wgp_stamps.log
  OUT=$O REPS=2 bash tests/tools/ab_run.sh prev "SAVP_LIB=$AB/libsavp_hip_wgprev.so" new ""
  echo "total $(( $(date +%s)-t0 ))s"
  ;;
*) echo "usage: r05_calls.sh <1|2|3|4|5|7|8|9|12|13|14|16|18|19|20|21>  (variant libraries under video_prediction_amd/ab/ are built first: tests/tools/build_variant.sh)"; exit 2 ;;
esac
