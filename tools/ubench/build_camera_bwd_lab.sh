#!/bin/bash
# Builds the three variants of tools/ubench/camera_bwd_lab.hip (gfx950): v0 the product's source as it is, v1 the noise-grid taps as three
# one-word loads each (a sed-made copy of the product's camera_rays.hip under /tmp), v2 the product's source with a warm-up touch of the grid,
# v4 the first two taps and the first blend recorded into the dump (columns fx..y),
# v3 the four taps loaded, then s_waitcnt vmcnt(0) and 32 idle cycles before their first use.
set -e
cd "$(dirname "$0")/../.."
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iscnerf_amd/csrc/device -Iscnerf_amd/csrc -Iinclude -Wall -Wno-unused-function -shared"
v=$(mktemp -d /tmp/camvar.XXXXXX)
python3 - "$v/camera_rays_split.hip" <<'P'
import sys
s = open("scnerf_amd/csrc/camera_rays.hip").read()
old = "    return v3(p[0], p[1], p[2]);\n"
assert s.count(old) == 1
new = ("    const float* p1 = p + 1;\n    const float* p2 = p + 2;\n"
       "    asm volatile(\"\" : \"+v\"(p1));\n    asm volatile(\"\" : \"+v\"(p2));     // adjacency hidden: three global_load_dword\n"
       "    return v3(*p, *p1, *p2);\n")
open(sys.argv[1], "w").write(s.replace(old, new))
P
python3 - "$v/camera_rays_settle.hip" <<'P'
import sys
s = open("scnerf_amd/csrc/camera_rays.hip").read()
old = """    const Vec3 a = t.wx0 * grid_at(g, gw, t.y0, t.x0) + t.wx1 * grid_at(g, gw, t.y0, t.x1);
    const Vec3 b = t.wx0 * grid_at(g, gw, t.y1, t.x0) + t.wx1 * grid_at(g, gw, t.y1, t.x1);
"""
assert s.count(old) == 1
new = """    Vec3 g00 = grid_at(g, gw, t.y0, t.x0), g01 = grid_at(g, gw, t.y0, t.x1), g10 = grid_at(g, gw, t.y1, t.x0), g11 = grid_at(g, gw, t.y1, t.x1);
    asm volatile("s_waitcnt vmcnt(0)\\n\\ts_nop 7\\n\\ts_nop 7\\n\\ts_nop 7\\n\\ts_nop 7"
                 : "+v"(g00.x), "+v"(g00.y), "+v"(g00.z), "+v"(g01.x), "+v"(g01.y), "+v"(g01.z), "+v"(g10.x), "+v"(g10.y), "+v"(g10.z),
                   "+v"(g11.x), "+v"(g11.y), "+v"(g11.z));
    const Vec3 a = t.wx0 * g00 + t.wx1 * g01;
    const Vec3 b = t.wx0 * g10 + t.wx1 * g11;
"""
open(sys.argv[1], "w").write(s.replace(old, new))
P
python3 - "$v/camera_rays_trace.hip" <<'P'
import sys
s = open("scnerf_amd/csrc/camera_rays.hip").read()
def once(old, new):
    global s
    assert s.count(old) == 1, old
    s = s.replace(old, new)
once("""__device__ __forceinline__ Vec3 sample_grid(const float* g, int gw, const Taps& t) {
    const Vec3 a = t.wx0 * grid_at(g, gw, t.y0, t.x0) + t.wx1 * grid_at(g, gw, t.y0, t.x1);
""", """__device__ __forceinline__ Vec3 sample_grid(const float* g, int gw, const Taps& t, float* dbg = nullptr) {
    const Vec3 g00 = grid_at(g, gw, t.y0, t.x0), g01 = grid_at(g, gw, t.y0, t.x1);
    const Vec3 a = t.wx0 * g00 + t.wx1 * g01;
    if (dbg) { dbg[0] = g00.x; dbg[1] = g00.y; dbg[2] = g00.z; dbg[3] = g01.y; dbg[4] = t.wx0; dbg[5] = a.y; }
""")
once("    float ux, uy, rx, ry, sx, sy;", "    float ux, uy, rx, ry, sx, sy;\n    float dbg[6];")
once("        r = r + a.scale_d * sample_grid(a.grid_d, a.gw, f->taps);", "        r = r + a.scale_d * sample_grid(a.grid_d, a.gw, f->taps, f->dbg);")
open(sys.argv[1], "w").write(s)
P
hipcc $FLAGS tools/ubench/camera_bwd_lab.hip -o tools/ubench/libcamera_bwd_lab_v0.so &
hipcc $FLAGS -DLAB_TRACE=1 -DLAB_CAMERA_SOURCE="\"$v/camera_rays_trace.hip\"" tools/ubench/camera_bwd_lab.hip -o tools/ubench/libcamera_bwd_lab_v4.so &
hipcc $FLAGS -DLAB_CAMERA_SOURCE="\"$v/camera_rays_settle.hip\"" tools/ubench/camera_bwd_lab.hip -o tools/ubench/libcamera_bwd_lab_v3.so &
hipcc $FLAGS -DLAB_CAMERA_SOURCE="\"$v/camera_rays_split.hip\"" tools/ubench/camera_bwd_lab.hip -o tools/ubench/libcamera_bwd_lab_v1.so &
hipcc $FLAGS -DLAB_WARM=1 tools/ubench/camera_bwd_lab.hip -o tools/ubench/libcamera_bwd_lab_v2.so &
wait
ls -la tools/ubench/libcamera_bwd_lab_v*.so
