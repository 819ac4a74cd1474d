#!/bin/bash
# profiles/rN_valu_mix.json: the static VALU mix of both tick-path rollout kernels (CPU only: hipcc cross-compiles)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); N=${1:-r4}
make -C $R/motion_planning_amd/csrc asm > /dev/null
python3 - <<PY
import json, subprocess, sys
R="$R"
def mix(*a): return json.loads(subprocess.check_output([sys.executable, R+"/tools/valu_mix.py", *a]))
out={"rollout_kernel": mix(),
     "rollout_pk_kernel": mix("--asm", R+"/build/asm/rollout_pk-hip-amdgcn-amd-amdhsa-gfx950.s", "--symbol", "rollout_pk_kernelILi1ELi4ELi0E", "--steps-per-iter", "12"),
     "rollout_fused_kernel": mix("--asm", R+"/build/asm/rollout_fused-hip-amdgcn-amd-amdhsa-gfx950.s", "--symbol", "rollout_fused_kernelILi4ELb0E", "--steps-per-iter", "6"),   # (the one-wave form: a pair of the split form issues the same instructions between its two waves)
     "rollout_pk_kernel_noise_packing_1": mix("--asm", R+"/build/asm/rollout_pk-hip-amdgcn-amd-amdhsa-gfx950.s", "--symbol", "rollout_pk_kernelILi1ELi4ELi1E", "--steps-per-iter", "16")}
json.dump(out, open(R+"/profiles/$N"+"_valu_mix.json","w"), indent=1, sort_keys=True)
for k,v in out.items(): print(k, "VALU/sample-step %.1f, issue cycles/sample-step %.1f" % (v["valu_per_step"], v["issue_cycles_per_step"]))
PY
