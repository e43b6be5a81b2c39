# GPU box: attribute the fused chain's time (profiling build in a scratch copy so the shipped .so stays clean)
set -e
export FCP_BUILD_PROFILING=1
python face-crop-plus_amd/build_native.py --force > /dev/null
for a in 0 1 2 3 4 8 12; do FCP_CHAIN_ABLATE=$a python tools/bench_chain.py; done
