set -e
TAG=${1:-r03_e}
R=/root/repo; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest $R/tests -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider > $O/gputests.log 2>&1
tail -2 $O/gputests.log
cd $R && timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err && cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -o s -- python3 $R/bench.py --steps 20 --no-cpu-baseline --no-train-wgan > $O/stats.log 2>&1
cp /tmp/p_stats/s_kernel_stats.csv $O/kernel_stats.csv
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -o f -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-train-wgan > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -o w -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-train-wgan > $O/write.log 2>&1
cd $R && python tools/pmc_traffic.py /tmp/p_fetch /tmp/p_write $TAG $O/kernel_stats.csv > $O/pmc.log 2>&1 && cp profiles/pmc_traffic.json profiles/${TAG}_pmc_fetch.csv profiles/${TAG}_pmc_write.csv $O/ && cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_hbm -o h -- python3 $R/tools/bench_hbm_kernels.py > $O/hbm.log 2>&1
cd $R && python tools/bench_hbm_kernels.py --from-stats /tmp/p_hbm/h_kernel_stats.csv $O/hbm_kernels_alone.json >> $O/hbm.log 2>&1
timeout -k 10 200 python bench.py --variant wgan_clip > $O/bench_wgan_clip.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --variant attention > $O/bench_attention.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --variant graph > $O/bench_graph.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --dtype bf16 --no-cpu-baseline --no-train-wgan > $O/bench_bf16.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --lengths adfa --no-cpu-baseline --no-train-wgan > $O/bench_adfa.json 2>> $O/bench.err
timeout -k 10 200 python bench.py --lengths adfa --gen-seq-len 120 --no-cpu-baseline --no-train-wgan > $O/bench_adfa_gen120.json 2>> $O/bench.err
echo ALLDONE
