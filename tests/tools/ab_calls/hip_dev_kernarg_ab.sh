cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "unset A=1" "dev1 HIP_FORCE_DEV_KERNARG=1" "dev0 HIP_FORCE_DEV_KERNARG=0" "unset2 A=1" "dev1b HIP_FORCE_DEV_KERNARG=1"; do
  set -- $v; name=$1; shift
  env "$@" python bench.py --steps 40 --warmup 3 --no-f32 --no-cpu-baseline --inst-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', 'ms/step %.2f'%d['ms_per_step'], d['config']['submission'])"
done
