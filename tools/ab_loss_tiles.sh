# tile shapes of the image-space loss kernels (gsr_loss.hip: GEO_TX x GEO_TY, SS_TY), same box, rebuilds of that one file
cd $GRAFT_REPO_ROOT
CM="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -munsafe-fp-atomics"
run() { echo "$1: $(python tools/bench_losses.py 2>&1 | tail -1)"; }
run "default (geo 64x4, ssim 32x30)"
for v in "-DGEO_TX=16 -DGEO_TY=16 -DSS_TY=32" "-DGEO_TX=32 -DGEO_TY=8 -DSS_TY=24" "-DGEO_TX=128 -DGEO_TY=2 -DSS_TY=16"; do
  touch gs-sr_amd/csrc/gsr_loss.hip
  make -C gs-sr_amd/csrc COMMON="$CM $v" > /tmp/mk.log 2>&1 || { echo "$v: build failed"; tail -3 /tmp/mk.log; continue; }
  run "$v"
done
touch gs-sr_amd/csrc/gsr_loss.hip; make -C gs-sr_amd/csrc > /dev/null 2>&1; run default_again
