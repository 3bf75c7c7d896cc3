cd $GRAFT_REPO_ROOT; R=$PWD; export TMPDIR=/tmp
for tag in nodense base; do
  LIBF=$R/rgbdslam_v2_amd/librgbdfe_$tag.so; [ $tag = base ] && LIBF=$R/rgbdslam_v2_amd/librgbdfe.so
  cd /tmp; rm -rf $R/gpurun_out/pmc_dense_$tag
  RGBDFE_LIB=$LIBF timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmc_dense_$tag -o p -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --depth-noise 0.002 > /dev/null 2>&1
  cd $R
  python - $R/gpurun_out/pmc_dense_$tag $tag <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda:collections.defaultdict(float)); n=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-28:]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    if r['Counter_Name']=='SQ_INSTS_LDS': n[k]+=1
for k in agg:
    if 'refine' in k or 'hyp' in k or 'select_ransac' in k or 'prep' in k:
        print(sys.argv[2], '%-30s launches %3d  bank-conflict cycles/launch %.3e  LDS insts/launch %.3e' % (k, n[k], agg[k]['SQ_LDS_BANK_CONFLICT']/max(n[k],1), agg[k]['SQ_INSTS_LDS']/max(n[k],1)))
PY
  find $R/gpurun_out/pmc_dense_$tag -name "*.db" -delete
done
