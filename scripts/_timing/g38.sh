cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
show() { tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['strong_scaling']; print(d['value'], s['pairs_per_s'], s['seconds'], s.get('pairs_per_s_max'), s.get('h2d_gbps'), {k: round(v,1) for k,v in s['pipeline_rank0'].items()})"; }
{
echo -n "no_secondary, cpu 512: "; timeout 400 python bench.py --no_secondary 2>&1 | show
echo -n "secondary, cpu 0: "; timeout 600 python bench.py --cpu_sample 0 2>&1 | show
echo -n "no_secondary, cpu 0: "; timeout 400 python bench.py --no_secondary --cpu_sample 0 2>&1 | show
} > gpurun_out/r04_g38_e2e_where.log 2>&1
exit 0
