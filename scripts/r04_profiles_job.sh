set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04prof; mkdir -p $O
timeout -k 10 420 python3 bench.py > $O/r04_bench_default_run.json 2> $O/bench_default.err; tail -c 300 $O/r04_bench_default_run.json; echo
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-public-api --no-alone --no-other-sources > $O/r04_bench_under_rocprof.json 2> $O/stats.err
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/r04_bench_kernel_stats.csv
python3 scripts/pipeline_timeline.py $O/stats > $O/r04_pipeline_timeline.txt; head -16 $O/r04_pipeline_timeline.txt
rm -rf $O/stats
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-public-api --no-alone --no-other-sources > /dev/null 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-public-api --no-alone --no-other-sources > /dev/null 2> $O/pmc_write.err
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write k_moran_score $O/pmc_out | cut -c1-600
rm -rf $O/pmc_fetch $O/pmc_write
SC_LIB=spatialcore_amd/libvar_phiprofile.so SC_PHI_PROFILE=1 timeout -k 10 100 python3 scripts/generator_probe.py 1000000 1000 > $O/r04_generator_chain_profile.txt 2>&1
SC_PHI_WAIT_LOG=1 SC_LIB=spatialcore_amd/libvar_phiprofile.so timeout -k 10 200 python3 scripts/pipeline_profile.py 500 3 >> $O/r04_generator_chain_profile.txt 2>&1
tail -5 $O/r04_generator_chain_profile.txt | cut -c1-300
