#!/bin/bash
# GPU box: joule ledger of the conv kernel classes.  For every build variant each class runs alone in
# a steady loop for $SECS s while rocm-smi samples socket power / sclk (tools/smi_loop.sh); J per launch = W x us.
#   bash tools/ledger.sh > gpurun_out/r4_ledger.txt
# Variants are experiment builds made in the box's scratch copy of the repo (wrong results by construction): the ablation
# switches are NOT part of the product kernels any more — tools/probes/experiment_switches_r04.patch re-adds them to the
# scratch copy (it is the reverse of the round-4 clean-up commit), and the product sources + library are restored at the end.
cd $GRAFT_REPO_ROOT
cp -r face-crop-plus_amd/csrc /tmp/csrc.product
patch -p1 -s < tools/probes/experiment_switches_r04.patch || { echo "patch failed"; exit 1; }
KERNELS=${KERNELS:-"chain pair2 pair3 big3x3 big1x1 dma1x1 dma3x3s2 wide halo stem copy"}
build() { env "$@" python face-crop-plus_amd/build_native.py --force > /tmp/ledger_build.log 2>&1 || { echo "BUILD FAILED: $*"; tail -5 /tmp/ledger_build.log; }; }
run() {   # label, env assignments...
  local label=$1; shift
  echo "== $label"
  env "$@" bash tools/smi_loop.sh "$KERNELS"
}
echo "idle: $(/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed -E 's/.*\(([0-9]+Mhz)\).*/\1/; s/.*Power \(W\): ([0-9.]+).*/\1 W/' | tr '\n' ' ')"
build FCP_X=0
run "full (product build)" FCP_X=0
build FCP_BUILD_FLAGS="-include tools/probes/fcp_no_mfma.h"
run "no MFMA (every f16 MFMA an empty asm statement; all data movement kept)" FCP_X=0
build FCP_BUILD_DEFINES="FCP_CONV_PROFILING"
run "profiling build, nothing ablated (control)" FCP_X=0
KERNELS="chain pair2 pair3" run "chain: no out stores" FCP_CHAIN_ABLATE=1
KERNELS="chain pair2 pair3" run "chain: no residual loads" FCP_CHAIN_ABLATE=2
KERNELS="chain pair2 pair3" run "chain: no out stores, no residual loads" FCP_CHAIN_ABLATE=3
KERNELS="chain pair2 pair3" run "chain: half of the filter DMA instructions" FCP_CHAIN_ABLATE=16
KERNELS="chain pair2 pair3" run "chain: no staging of the chunk tile (epilogue LDS round trip)" FCP_CHAIN_ABLATE=32
KERNELS="big3x3 big1x1 dma1x1 dma3x3s2 wide halo" run "conv kernels: no output stores" FCP_CONV_ABLATE=64
build FCP_BUILD_DEFINES="FCP_BIG_ABLATE=1"
KERNELS="big3x3 big1x1" run "256-row: no operand DMA after the prologue (stale operands: MFMAs keep real data)" FCP_X=0
build FCP_BUILD_DEFINES="FCP_BIG_ABLATE=16"
KERNELS="big3x3 big1x1" run "256-row: no fragment reads in the loop" FCP_X=0
build FCP_BUILD_DEFINES="FCP_BIG_ABLATE=17"
KERNELS="big3x3 big1x1" run "256-row: MFMAs only (no DMA, no fragment reads)" FCP_X=0
build FCP_X=0
rm -rf face-crop-plus_amd/csrc && cp -r /tmp/csrc.product face-crop-plus_amd/csrc   # product sources and library back
