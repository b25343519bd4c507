#!/bin/bash
# tools/_bin/chol_persist_stamps: the library's objects with linalg.hip rebuilt under -DHB_CP_STAMPS + the stamp reader
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
python3 -c "import sys; sys.path.insert(0, '$root'); from henbun_amd import _build; _build.build()"
mkdir -p "$root/tools/_bin"
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-result"
hipcc $F -DHB_CP_STAMPS -c "$root/henbun_amd/csrc/linalg.hip" -o "$root/tools/_bin/linalg_stamps.o"
hipcc $F -c "$root/tools/chol_persist_stamps.hip" -o "$root/tools/_bin/chol_persist_stamps.o"
objs=$(ls "$root"/henbun_amd/csrc/_obj/*.o | grep -v linalg.o)
hipcc --offload-arch=gfx950 -o "$root/tools/_bin/chol_persist_stamps" "$root/tools/_bin/chol_persist_stamps.o" "$root/tools/_bin/linalg_stamps.o" $objs -ldl
