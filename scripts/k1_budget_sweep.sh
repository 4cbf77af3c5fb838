set -e
cd /root/repo
for b in 12288 16384 24576 32768 49152; do
  rm -f deep_interpolation_clustering_amd/csrc/dic_interp.o
  make -s -C deep_interpolation_clustering_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$PWD/include -Wall -Wno-unused-function -DDIC_K1_LDS_BUDGET=$b" > /dev/null 2>&1
  echo "== LDS budget $b"; python scripts/kbench.py 32768 10 2>/dev/null | grep "sci_cci_fwd"
done
rm -f deep_interpolation_clustering_amd/csrc/dic_interp.o; make -s -C deep_interpolation_clustering_amd/csrc > /dev/null 2>&1
