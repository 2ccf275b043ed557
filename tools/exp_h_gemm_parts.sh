#!/bin/bash
# What the 256 x 256 bf16 GEMM (c1b_gemm_ps_kernel) spends where: timing-only builds of csrc/conv1d_bf16.hip
#   for v in NOEPI NOMFMA NODMA NOSTORE; do python -m asvspoof2021_air_amd.build --variant g2$v conv1d_bf16.hip -DG2_X_$v; done
# (results of these builds are garbage).  Usage (GPU box): bash tools/exp_h_gemm_parts.sh
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/asvspoof2021_air_amd/_lib
for V in "" g2NOEPI g2NOMFMA g2NODMA g2NOSTORE; do
  if [ -z "$V" ]; then unset AIR_HIP_LIB; echo "== product"; else export AIR_HIP_LIB=$L/libair_hip.$V.so; echo "== $V"; fi
  python tools/kbench_h_gemm.py 2>&1 | grep -v libdrm
done
