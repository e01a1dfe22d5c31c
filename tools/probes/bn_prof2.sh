# bn_stats / bn_bwd_reduce per-kernel time vs workgroup count, with and without the final global atomics (experiment build)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$ROOT
rm -f $OUT/r4h_bn_prof.txt
for LIB in default noatomic; do
  if [ $LIB = noatomic ]; then export ITG_LIB=$ROOT/ab_libs/libitg_noatomic.so; else unset ITG_LIB; fi
  for B in 128 256 512 1024; do
    export ITG_BN_RED_BLOCKS=$B
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4h_tmp -- python3 $ROOT/tools/membound_bench.py bn13 > $OUT/r4h_tmp.log 2>&1
    echo "== lib $LIB blocks $B" >> $OUT/r4h_bn_prof.txt
    python3 $ROOT/tools/kstats.py $OUT/r4h_tmp bn_ >> $OUT/r4h_bn_prof.txt
    rm -rf $OUT/r4h_tmp
  done
done
cat $OUT/r4h_bn_prof.txt
