mkdir -p gpurun_out/r06j
for m in tsdf default; do
for k in 1 2 3 4 5; do echo "== $m exit at $k"; VCY_HIP_LIB=$PWD/build/variants/exit$k/libvacancy_hip.so bash profiles/tools/pmc_single_view.sh gpurun_out/r06j/sv_${m}_$k $m 2>&1 | grep "view [03]"; done
echo "== $m full"; VCY_HIP_LIB=$PWD/build/variants/one/libvacancy_hip.so bash profiles/tools/pmc_single_view.sh gpurun_out/r06j/sv_${m}_full $m 2>&1 | grep "view [03]"
done | tee gpurun_out/r06j/attribution.txt
