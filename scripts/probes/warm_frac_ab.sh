cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in 32:2 64:2 48:2 96:2 64:2 32:2; do
  MUON_AMD_LSI_WARM=$spec timeout 600 python bench.py --steps 6 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$spec', round(d['ms_per_step'],2), d['config']['lsi']['spmm_per_step'], d['config']['lsi']['warm_start'], d['config']['lsi']['lanczos_bounds'], d['config']['lsi']['angle_bound'])"
done
