#!/bin/bash
# Runs ON THE GPU BOX: the DPP-mirror MFCC variant against the shipped one -- parity tests on the variant, then the interleaved A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r3mfccab; mkdir -p $O
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-result -Wno-pass-failed"
cp rustpotter_amd/librustpotter_hip.so $O/keep.so
touch rustpotter_amd/csrc/rp_mfcc.hip
make -C rustpotter_amd/csrc -j8 CXXFLAGS="$BASE -DRP_MFCC_DPP_MIRROR" > $O/make_dpp.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q -k "mfcc or golden or smoke" 2>&1 | tail -5
timeout 600 python tests/sweep_parity.py --cases 0 --mfcc-cases 400 2>&1 | tail -2
cp $O/keep.so rustpotter_amd/librustpotter_hip.so
bash tools/r3_mfcc_ab.sh "" "-DRP_MFCC_DPP_MIRROR"
