#!/bin/bash
# tools/_bin/mlp_stamps: the library's objects with mlp.hip rebuilt under -DHB_MLP_STAMPS + the stamp reader
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
python3 -c "import sys; sys.path.insert(0, '$root'); from henbun_amd import _build; _build.build()"
mkdir -p "$root/tools/_bin"
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result"
hipcc $F -DHB_MLP_STAMPS -c "$root/henbun_amd/csrc/mlp.hip" -o "$root/tools/_bin/mlp_stamps_k.o"
hipcc $F -c "$root/tools/mlp_stamps.hip" -o "$root/tools/_bin/mlp_stamps.o"
objs=$(ls "$root"/henbun_amd/csrc/_obj/*.o | grep -v "/mlp.o")
hipcc --offload-arch=gfx950 -o "$root/tools/_bin/mlp_stamps" "$root/tools/_bin/mlp_stamps.o" "$root/tools/_bin/mlp_stamps_k.o" $objs -ldl
