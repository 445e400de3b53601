REPO=$GRAFT_REPO_ROOT; OUT=$REPO/gpurun_out/prof_tmp; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq -- python3 $REPO/bench.py --steps 4 --warmup 1 --cpu-seconds 0 --no-kernel-timing --frames-per-step 256 > $OUT/b.json 2> $OUT/log.txt
python3 - <<'PY'
import pandas as pd, glob, os
f = glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof_tmp/pmc_sq/*/*_counter_collection.csv')[0]
df = pd.read_csv(f); df = df[df.Kernel_Name.str.contains('k_feature|k_project')]
df['k'] = df.Kernel_Name.str.extract(r'(k_[a-z_]+)')
print(df.groupby(['k','Counter_Name']).Counter_Value.mean().unstack(0))
print(df[['VGPR_Count','SGPR_Count','LDS_Block_Size']].drop_duplicates())
PY
