#!/bin/bash
# Rebuild only the named translation units and relink (kernel experiments; the other units stay self-consistent as long as
# dvn_internal.h's structs are unchanged).  `make` afterwards rebuilds everything that depends on the edited headers.
# usage: ./quick_tu.sh dvn_ntt [dvn_msm_g1 ...]
set -e
cd "$(dirname "$0")"
for tu in "$@"; do
  hipcc -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops --offload-arch=gfx950 -fPIC -Wno-unused-result -c $tu.hip -o $tu.o
done
hipcc --offload-arch=gfx950 -shared -fPIC -o libdvn_prover.so dvn_core.o dvn_msm_g1.o dvn_msm_g2.o dvn_ntt.o dvn_groth16.o dvn_prover_pkg.o dvn_bls12377.o dvn_bw6761.o dvn_bls12381.o dvn_artifacts.o dvn_comm.o
