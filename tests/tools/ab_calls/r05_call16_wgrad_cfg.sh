O=gpurun_out/r05p; mkdir -p $O
for cfg in 0 48 84 44; do SAVP_WGP_CFG=$cfg BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg$cfg /"; done | tee $O/wgrad_cfg_sweep.log
for sp in 256 512 768; do SAVP_WGP_SPLIT=$sp BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/split$sp /"; done | tee -a $O/wgrad_cfg_sweep.log
for sp in 512 768 1024; do SAVP_WGP_CFG=48 SAVP_WGP_SPLIT=$sp BF16=1 NOBIAS=1 python tests/tools/bench_wgrad.py lstm_h0 lstm_h1 lstm_h2 2>&1 | grep -v amdgpu.ids | sed "s/^/cfg48_split$sp /"; done | tee -a $O/wgrad_cfg_sweep.log
