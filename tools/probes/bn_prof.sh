# per-kernel durations of the BatchNorm benchmark rows (rocprofv3 --stats), old vs new norm.hip
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
for V in old new; do
  if [ $V = old ]; then export ITG_LIB=$ROOT/ab_libs/libitg_oldnorm.so; else unset ITG_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4g_bn_$V -- python3 $ROOT/tools/membound_bench.py bn13 > $OUT/r4g_bn_$V.log 2>&1
  echo "== $V" >> $OUT/r4g_bn_prof.txt
  python3 $ROOT/tools/kstats.py $OUT/r4g_bn_$V >> $OUT/r4g_bn_prof.txt
  rm -rf $OUT/r4g_bn_$V
done
cat $OUT/r4g_bn_prof.txt
