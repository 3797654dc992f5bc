# Repro attempt for the round-1 observation "instances that spill more than ~256 B per lane compute wrong values":
# every step-kernel instance compiled for 5 waves per SIMD (96 VGPRs; scratch per lane in build/res_occ5all.log), run
# through the per-slot allocation-trace parity tests and the soak.  Build: hipcc ... -DRS_OCC_OTHER=5 -o build/libranslice_occ5all.so
for lib in libranslice_occ5all.so; do
echo "== $lib"
RANSLICE_LIB=network-slicing_amd/csrc/build/$lib timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py "tests/test_gpu_fullsize.py::test_soak_every_replica" -q -m gpu 2>&1 | tail -4
done
