#!/bin/bash
# Round 4, config 5 (beam model) artefacts on the final tree: kernel trace, counters, 8 cycles with and without the sector windows, walk statistics.
set -u
mkdir -p gpurun_out/prof gpurun_out/r04
export TMPDIR=/tmp CONFIG5_ALL_STEPS=1
O=$GRAFT_REPO_ROOT/gpurun_out/r04
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
{ for v in 1 0 1 0; do echo "beam_sectors=$v"; BELUGA_MCL_BEAM_SECTORS=$v python tools/bench_configs.py 5 --steps 8 2>/dev/null; done; } > $O/config5.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof/c5 -o trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 3 > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/c5/trace_results.db | head -12 > $O/kernel_trace_config5.txt
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/c5
i=0
echo "# rocprofv3 --kernel-trace --pmc <group> of: tools/bench_configs.py 5 --steps 2 (BeamSensorModel, 1M x 1080, sector windows)" > $O/pmc_config5.txt
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $pmc -d $GRAFT_REPO_ROOT/gpurun_out/prof/b$i -o pmc -- python $GRAFT_REPO_ROOT/tools/bench_configs.py 5 --steps 2 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/prof/b$i.err || echo "beam pass $i failed"
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $GRAFT_REPO_ROOT/gpurun_out/prof/b$i/pmc_results.db pmc 2>/dev/null | grep "^PMC" | grep -E "reweight_beam" >> $O/pmc_config5.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof/b$i
done
cd $GRAFT_REPO_ROOT
bash tools/gpu_beam_ablate.sh 1 2 > $O/beam_ablations.txt 2>&1
bash tools/gpu_beam_stats.sh > /dev/null 2>&1; cp gpurun_out/beam_stats.txt $O/beam_walk_stats.txt
