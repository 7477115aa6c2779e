cd "${GRAFT_REPO_ROOT:-/root/repo}"
# larger slices / more power steps of the warm start at c3: can the main process do with fewer than 5 full products?
for spec in 32:2 8:2 8:3 4:3 16:3 16:4; do
  MUON_AMD_LSI_WARM=$spec timeout 600 python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$spec', round(d['ms_per_step'],2), d['config']['lsi']['spmm_per_step'], d['config']['lsi']['warm_start'], d['config']['lsi']['lanczos_bounds'], d['config']['lsi']['angle_bound'], d['config'].get('parity_lsi_angle_rad'))"
done
