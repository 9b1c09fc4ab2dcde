# The other BASELINE configurations at size + the generator's correctness sweeps on the final r04 build (GPU box).
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04prof; mkdir -p $O
timeout -k 10 200 python3 scripts/lee_scale_probe.py 10 10 > $O/r04_lee_config2_10x10_pairs.json 2>$O/lee.err; cut -c1-400 $O/r04_lee_config2_10x10_pairs.json
timeout -k 10 200 python3 scripts/lee_scale_probe.py 100 100 199 shared > $O/r04_lee_config2_shared_permutations.json 2>>$O/lee.err; cut -c1-300 $O/r04_lee_config2_shared_permutations.json
timeout -k 10 300 python3 bench.py --config 3 --genes 250 --steps 3 --warmup 1 > $O/r04_config3_share_one_gpu.json 2>$O/c3.err; python3 -c "
import json; d=json.loads(open('$O/r04_config3_share_one_gpu.json').read().strip().splitlines()[-1]); print('config3 share', d['value'], d['ms_per_step'], d['strong_scaling'])"
timeout -k 10 400 python3 bench.py --config 3 --steps 1 --warmup 1 > $O/r04_config3_all_2000_genes_one_gpu.json 2>>$O/c3.err; python3 -c "
import json; d=json.loads(open('$O/r04_config3_all_2000_genes_one_gpu.json').read().strip().splitlines()[-1]); print('config3 full', d['value'], d['ms_per_step'], d['strong_scaling'])"
timeout -k 10 200 python3 scripts/config_scale_probe.py local > $O/r04_local_configs.json 2>$O/local.err; cut -c1-400 $O/r04_local_configs.json
timeout -k 10 200 python3 scripts/config_scale_probe.py enrich 10000 philox > $O/r04_config4_enrichment_philox.json 2>>$O/local.err; cut -c1-300 $O/r04_config4_enrichment_philox.json
timeout -k 10 200 python3 scripts/config_scale_probe.py enrich 10000 numpy > $O/r04_config4_enrichment_numpy.json 2>>$O/local.err; cut -c1-300 $O/r04_config4_enrichment_numpy.json
timeout -k 10 300 python3 scripts/generator_sweep.py > $O/r04_generator_sweep.txt 2>&1; tail -2 $O/r04_generator_sweep.txt
timeout -k 10 400 python3 scripts/pipeline_soak.py 500 100 2>&1 | cut -c1-200 > $O/r04_pipeline_soak_100_reps.txt; tail -2 $O/r04_pipeline_soak_100_reps.txt
