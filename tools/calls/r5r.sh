mkdir -p gpurun_out/r5r
timeout 1500 python bench.py --steps 4 --warmup 1 --ops-json gpurun_out/r5r/ops_b192.json > gpurun_out/r5r/bench_b192.json 2> gpurun_out/r5r/bench_b192.err; tail -c 300 gpurun_out/r5r/bench_b192.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5r/bench_b192.json'))
print({k:v for k,v in d.items() if k not in ('roofline','config','cpu_baseline')})
c=d['config']; print({k:c[k] for k in ('latency_1scene_s','vae','vae_decode_ms_per_scene','scenes_per_s_incl_vae_decode','full_cond_scenes_per_s','mfma_frac_end_to_end')}, c['hires']['scenes_per_s'])
r=d['roofline']; print({k:v for k,v in r.items() if k not in ('per_kernel','per_family','traffic')})
PY
