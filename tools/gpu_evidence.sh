#!/bin/bash
# Runs ON the GPU box (through gpurun): everything the round's profiles/ entries are generated from.
#   gpurun --timeout 2400 -- 'bash tools/gpu_evidence.sh'
# then locally:  python tools/collect_evidence.py
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/evidence
rm -rf $O; mkdir -p $O
cd $R
export LAS_ROUND=r06
rm -f gpurun_out/parity_observed.jsonl
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o x -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary > $O/stats.log 2>&1
F="--steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_step/fetch -o x -- python3 $R/bench.py $F > $O/pmc_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_step/write -o x -- python3 $R/bench.py $F > $O/pmc_step_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o x -- python3 $R/tools/ubench_rec.py > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o x -- python3 $R/tools/ubench_rec.py > $O/pmc_write.log 2>&1
# BASELINE configs[4] (B = 8, T = 3000): kernel stats of the step, HBM counters of the layer-0 recurrence (T_l = 1500) and of the decode kernels
# (T' = 375), LDS counters of the decode kernels (P: keys resident; S at this T: keys do not fit, per-step kernels)
FL="--workload P_long --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary"
rocprofv3 --kernel-trace --stats -d $O/stats_long -o x -- python3 $R/bench.py $FL > $O/stats_long.log 2>&1
export B=8 T=1500
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_long_fetch -o x -- python3 $R/tools/ubench_rec.py > $O/pmc_long_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_long_write -o x -- python3 $R/tools/ubench_rec.py > $O/pmc_long_write.log 2>&1
unset B T
FL4="--workload P_long --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary"
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_long_step/fetch -o x -- python3 $R/bench.py $FL4 > $O/pmc_long_step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_long_step/write -o x -- python3 $R/bench.py $FL4 > $O/pmc_long_step_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY -d $O/pmc_long_lds -o x -- python3 $R/bench.py $FL4 > $O/pmc_long_lds.log 2>&1
FS4="--workload S_long --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary"
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY -d $O/pmc_slong_lds -o x -- python3 $R/bench.py $FS4 > $O/pmc_slong_lds.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/stats_slong -o x -- python3 $R/bench.py $FS4 > $O/stats_slong.log 2>&1
python3 $R/tools/lds_long_table.py $O > $O/lds_long.log 2>&1
# ordered timelines of one step (headline and configs[4]; the latter shows the two queues of the deferred weight-gradient groups)
python3 $R/tools/step_timeline.py $(find $O/stats -name "*.db" | head -1) > $O/step_timeline.log 2>&1
python3 $R/tools/step_timeline.py $(find $O/stats_long -name "*.db" | head -1) > $O/step_timeline_long.log 2>&1
# ... and the same step with the deferred weight-gradient groups switched ON (LAS_DEFER_DW=1; off by default): the two-queue trace
export LAS_DEFER_DW=1
rocprofv3 --kernel-trace -d $O/kt_long_defer -o x -- python3 $R/bench.py --workload P_long --batch 8 --steps 6 --warmup 3 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary > $O/kt_long_defer.log 2>&1
unset LAS_DEFER_DW
python3 $R/tools/step_timeline.py $(find $O/kt_long_defer -name "*.db" | head -1) > $O/step_timeline_long_deferred.log 2>&1
# MFMA-busy / stall / LDS counters of the GEMM in both arithmetic modes (1 = split-operand bf16 MFMA, the default; 0 = fp32 MFMA)
for a in 1 0; do
  export LAS_GEMM_ARITH=$a
  if [ $a = 1 ]; then MOPS=SQ_INSTS_VALU_MFMA_MOPS_BF16; else MOPS=SQ_INSTS_VALU_MFMA_MOPS_F32; fi
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY $MOPS GRBM_GUI_ACTIVE -d $O/pmc_gemm_a$a -o x -- python3 $R/tools/ubench_gemm_pmc.py > $O/pmc_gemm_a$a.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc_gemm_b$a -o x -- python3 $R/tools/ubench_gemm_pmc.py > $O/pmc_gemm_b$a.log 2>&1
done
unset LAS_GEMM_ARITH
# the matrix-pipe recurrences at B = 512: MFMA-busy / wait / LDS counters and HBM bytes (four separate passes)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE -d $O/pmc_recm_a -o x -- python3 $R/tools/ubench_rec_mfma_pmc.py > $O/pmc_recm_a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/pmc_recm_b -o x -- python3 $R/tools/ubench_rec_mfma_pmc.py > $O/pmc_recm_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_recm_f -o x -- python3 $R/tools/ubench_rec_mfma_pmc.py > $O/pmc_recm_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_recm_w -o x -- python3 $R/tools/ubench_rec_mfma_pmc.py > $O/pmc_recm_w.log 2>&1
cd $R
export LAS_ROUND=r06
TRACE=1 BS=128,512,768 python tools/ubench_rec_mfma.py > $O/rec_mfma_trace.log 2>&1
python tools/ubench_gemm_skf.py 32 > $O/gemm_skf_b32.log 2>&1
python tools/ubench_gemm_skf.py 128 > $O/gemm_skf_b128.log 2>&1
./tools/_bin/spin_timeout > $O/spin_timeout.log 2>&1
# the one-launch decode kernels of the reference's YAML sizes against the per-step chains: parity, times, phase traces (tools/check_big.py)
python tools/check_big.py > $O/big_decode.log 2>&1
python tools/ubench_rec_sweep.py > $O/rec_sweep.log 2>&1
python tools/ubench_gemm_split.py > $O/gemm_split.log 2>&1
python tools/ubench_gemm_big.py > $O/gemm_big.log 2>&1
python tools/ubench_gemm_planes.py > $O/gemm_planes.log 2>&1
# round 5: multi-head attention on the one-launch kernels against the per-step chains; the caller's step (solver.batch_iterator); the device LER
python tools/ubench_multihead.py 2>&1 | grep -v amdgpu.ids > $O/multihead.log
B=16 python tools/ubench_multihead.py 2>&1 | grep -v amdgpu.ids >> $O/multihead.log
python tools/solver_step_profile.py 300 2>&1 | grep -v amdgpu.ids > $O/solver_step.log
python tools/soak_mixed.py 1500 2>&1 | grep -v amdgpu.ids > $O/soak_mixed.log
(python tools/ubench_ler.py; B=7 U=600 python tools/ubench_ler.py; B=3 U=4095 python tools/ubench_ler.py) 2>&1 | grep "las_letter_error_rate" > $O/ler.log
# round 6: in-process A/B of this round's switches (tools/ab_step_option.py alternates the two values on one box) and the block -> XCD probe
(echo "# REC_EPOCH_SCRATCH (hand-off granules in the library's epoch-tagged scratch vs a fill per launch), headline step";
 python tools/ab_step_option.py REC_EPOCH_SCRATCH 1 0 60 4;
 echo "# KEYS_SPLITK (psi keys GEMM split over K + activation pass vs one pass), headline step";
 python tools/ab_step_option.py KEYS_SPLITK 1 0 60 4;
 echo "# DW_CONCURRENT (weight-gradient group beside dX on a second stream, joined in the call; runs drawn), headline step";
 python tools/ab_step_option.py DW_CONCURRENT 1024 0 60 4) 2>&1 | grep -v amdgpu.ids > $O/ab_switches.log
(echo "# DEFER_DW at BASELINE configs[4] (B = 8, T = 3000): weight-gradient groups on the side stream, XCD partition";
 B=8 T=3000 python tools/ab_step_option.py DEFER_DW 1 0 20 4;
 echo "# block -> XCD placement of the two partitioned launches (tools/xcd_probe.py)";
 B=8 T=3000 python tools/xcd_probe.py;
 echo "# recurrence + GEMM group on two streams, B = 8, T_l = 1500 (tools/ubench_overlap.py; unpartitioned launches)";
 B=8 T=1500 python tools/ubench_overlap.py) 2>&1 | grep -v amdgpu.ids > $O/defer_dw.log
# soak: 1500 consecutive training steps (~1e7 inter-workgroup hand-offs) must end without a device error word
python bench.py --steps 1500 --warmup 5 --no-cpu-baseline --no-sweep --no-mfma --no-roofline --no-secondary > $O/soak.json 2> $O/soak.err
for w in "P_long 8" "S_long 8" "S_train 32" "Y_train 16" "P_fwd 32" "S_fwd 32" "P_train 128"; do set -- $w; python bench.py --workload $1 --batch $2 --steps 10 --warmup 3 --no-cpu-baseline --no-sweep --no-mfma --no-secondary --no-roofline 2>/dev/null | tail -1; done > $O/variants.jsonl
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log; tail -c 300 $O/bench.json
# summaries are made HERE (the rocprofv3 databases are too large to travel: gpurun merges at most 64 MiB back) and copied next to the logs
LAS_ROUND=r06 python tools/collect_evidence.py > $O/collect.log 2>&1
mkdir -p $R/gpurun_out/r06_profiles && cp $R/profiles/r06_* $R/gpurun_out/r06_profiles/ 2>/dev/null
find $O -name "*.db" -delete
find $O -name "*.csv" -size +1M -delete
du -sh $R/gpurun_out
tail -3 $O/collect.log
