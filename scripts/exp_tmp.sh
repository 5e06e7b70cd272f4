cd $GRAFT_REPO_ROOT
O=gpurun_out/exp4; rm -rf $O; mkdir -p $O
run() { for K in 24 28 31 32 33 36 40 48; do python scripts/kbench.py --N 4000000 --D 12 --K $K 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin); print('D=%d K=%d logpdf %.4f (%.4f) ps/pair %.2f  resp %.4f  [$1]' % (r['D'], r['K'], r['logpdf']['ms'], r['logpdf']['ms_median'], r['logpdf']['ms']*1e9/(4e6*r['K']), r['vb_resp_only']['ms']))"; done; }
run default | tee $O/d12.txt
for rb in 30000 36000; do
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DPMC_D=12 -DPMC_PADDED=0 -DPMC_RESIDENT_BYTES=$rb -c pypmc_amd/csrc/pmc_persample.hip -o pypmc_amd/csrc/build/pmc_persample_d12_p0.o
hipcc --offload-arch=gfx950 -shared -fPIC -o pypmc_amd/lib/libpmc_hip.so pypmc_amd/csrc/build/*.o -ldl
run RESIDENT_BYTES=$rb | tee -a $O/d12.txt
done
