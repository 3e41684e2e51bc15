#!/usr/bin/env python3
"""Builds the DIAGNOSTIC variant libkmc_wstamps.so that tools/walk_stamps.py reads: the walk kernel with
s_memrealtime stamps per wave (start, LDS init done, first tile loaded / stepped, tile loop done, end)
in a __device__ array of its own plus kmc_debug_walk_stamps() to fetch them.  The product library
never contains any of this."""
import os, subprocess, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s=open(os.path.join(ROOT, 'k-mer-count_amd', 'csrc', 'kmc_walk.hip.h')).read()
def rep(old,new):
    global s
    assert old in s, old[:60]; s=s.replace(old,new,1)
rep('template <int KW, bool CANON>\n__global__ __launch_bounds__(KMC_WALK_THREADS)\nvoid kmc_walk_kernel(','__device__ unsigned long long g_walk_stamps[256 * 16 * 8];\ntemplate <int KW, bool CANON>\n__global__ __launch_bounds__(KMC_WALK_THREADS)\nvoid kmc_walk_kernel(')
rep('''    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);

    const bool warm''','''    const u64 mask_hi = kb <= 64 ? 0ull : ((1ull << (kb - 64)) - 1);
    unsigned long long* stp = &g_walk_stamps[((size_t)blockIdx.x * 16 + (tid >> 6)) * 8];
    if (lane == 0) stp[0] = __builtin_amdgcn_s_memrealtime();

    const bool warm''')
rep('''    const u32 s_root = root_id << 4;
''','''    const u32 s_root = root_id << 4;
    if (lane == 0) stp[1] = __builtin_amdgcn_s_memrealtime();
    u32 stamp_tiles = 0;
''')
rep('''        if (last) {
            step_phase(cur, tile);''','''        if (last) {
            if (lane == 0 && stamp_tiles == 0) stp[2] = __builtin_amdgcn_s_memrealtime();
            step_phase(cur, tile);
            if (lane == 0 && stamp_tiles == 0) stp[3] = __builtin_amdgcn_s_memrealtime();
            stamp_tiles++;''')
rep('''    nk = wave_sum_u64(nk);
    ndirect = wave_sum_u64(ndirect);
    if (lane == 0) {
        if (nk) atomicAdd''','''    if (lane == 0) { stp[4] = __builtin_amdgcn_s_memrealtime(); stp[6] = stamp_tiles; }
    nk = wave_sum_u64(nk);
    ndirect = wave_sum_u64(ndirect);
    if (lane == 0) {
        if (nk) atomicAdd''')
rep('''        if (tid == 0) { memo_out->nedges = L.nedges; memo_out->nnodes = L.nnodes; memo_out->tag = KMC_WALK_MEMO_TAG | (u64)k; }
    }
}''','''        if (tid == 0) { memo_out->nedges = L.nedges; memo_out->nnodes = L.nnodes; memo_out->tag = KMC_WALK_MEMO_TAG | (u64)k; }
    }
    if (lane == 0) stp[5] = __builtin_amdgcn_s_memrealtime();
}''')
walk_src = s
src=open(os.path.join(ROOT, 'k-mer-count_amd', 'csrc', 'kmc_api.hip')).read()
src+='''
extern "C" int kmc_debug_walk_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_stamps), sizeof(unsigned long long) * 256 * 16 * 8) == hipSuccess ? 0 : -3;
}
'''
w = tempfile.mkdtemp(prefix='kmc_wstamps_')
os.makedirs(os.path.join(w, 'pkg', 'csrc')); os.makedirs(os.path.join(w, 'include'))
for f in os.listdir(os.path.join(ROOT, 'k-mer-count_amd', 'csrc')):
    shutil.copy(os.path.join(ROOT, 'k-mer-count_amd', 'csrc', f), os.path.join(w, 'pkg', 'csrc', f))
shutil.copy(os.path.join(ROOT, 'include', 'kmc.h'), os.path.join(w, 'include', 'kmc.h'))
open(os.path.join(w, 'pkg', 'csrc', 'kmc_walk.hip.h'), 'w').write(walk_src)
open(os.path.join(w, 'pkg', 'csrc', 'kmc_api.hip'), 'w').write(src)
out = os.path.join(ROOT, 'k-mer-count_amd', 'libkmc_wstamps.so')
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-function', '-I../include', '-c', 'csrc/kmc_api.hip', '-o', 'kmc_api.o'], cwd=os.path.join(w, 'pkg'), check=True)
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-pthread', '-o', out, 'kmc_api.o', os.path.join(ROOT, 'k-mer-count_amd', 'kmc_host.o')], cwd=os.path.join(w, 'pkg'), check=True)
print(out)
