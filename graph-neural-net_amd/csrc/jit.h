// jit.h -- run-time specialisation of middle4_kernel for the net at hand (hiprtc).
//
// The hand-written kernel in middle4_kernel.h is a template over a shape policy.  With
// StaticShape<d0,...> every extent, LDS offset and K split is a compile-time constant, which at
// this problem size is worth 1.5x (10.6 vs 16 us for 784-300-100-10): a workgroup's critical
// path is a few thousand instructions, and runtime extents mean kernarg loads, integer divisions
// and loops with runtime bounds on that path.  Instead of enumerating shapes ahead of time, the
// library keeps the kernel SOURCES (embedded at build time, _generated/embedded_sources.h) and instantiates
// the same template for the caller's layer sizes with hiprtc -- the MI355X-native replacement for
// a tracing compiler: one explicit template instantiation, not a graph capture.
// Failure at any point (no hiprtc, compile error, load error) is not an error of the path: the
// caller keeps the ahead-of-time RuntimeShape kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "_generated/embedded_sources.h" // written by build.py from the kernel headers

namespace gnn {
namespace jit {

struct Specialised {
    hipModule_t module = nullptr;
    hipFunction_t fn[3] = {nullptr, nullptr, nullptr}; // forward only, forward + backward, the same with A_1 from K slabs
    int n_fn = 2;
    std::string log;
};

inline std::string shape_list(const int *dims, int L) {
    std::string s;
    for (int i = 0; i < L; i++) s += (i ? ", " : "") + std::to_string(dims[i]);
    return s;
}

// Compiles middle4_kernel<StaticShape<dims...>, act, outk, backward, false, slabs> for gfx950:
// (false, false), (true, false) and, when `with_slabs`, (true, true).
// Returns the code object (empty on failure; *log holds the compiler output).
// bf16: ONE kernel, (true, true, BF16 = true) -- the training kernel of the two-launch path in GNN_DTYPE_BF16 -- in slot 2.
inline std::vector<char> compile_middle4(const int *dims, int L, int act, int outk, bool with_slabs, bool bf16, std::string names[3], std::string *log) {
    std::vector<char> code;
    const std::string shape = "gnn::StaticShape<" + shape_list(dims, L) + ">";
    const int n_fn = with_slabs ? 3 : 2, b_first = bf16 ? 2 : 0;
    std::string expr[3];
    for (int b = b_first; b < n_fn; b++)
        expr[b] = "gnn::middle4_kernel<" + shape + ", " + std::to_string(act) + ", " + std::to_string(outk) + ", " +
                  (b ? "true" : "false") + ", false, " + (b == 2 ? "true" : "false") + ", " + (bf16 ? "true" : "false") + ">";
    const std::string src = "#include \"middle4_kernel.h\"\n";
    const char *hdr_src[] = {kEmbedded_kernels_h, kEmbedded_fused_kernels_h, kEmbedded_middle4_kernel_h};
    const char *hdr_name[] = {"kernels.h", "fused_kernels.h", "middle4_kernel.h"};
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "gnn_mid4_jit.hip", 3, hdr_src, hdr_name) != HIPRTC_SUCCESS) {
        if (log) *log = "hiprtcCreateProgram failed";
        return code;
    }
    for (int b = b_first; b < n_fn; b++) (void)hiprtcAddNameExpression(prog, expr[b].c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
    const hiprtcResult rc = hiprtcCompileProgram(prog, 3, opts);
    size_t ls = 0;
    if (log && hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        log->assign(ls, '\0');
        (void)hiprtcGetProgramLog(prog, &(*log)[0]);
    }
    if (rc == HIPRTC_SUCCESS) {
        bool ok = true;
        for (int b = b_first; b < n_fn && ok; b++) {
            const char *low = nullptr;
            ok = hiprtcGetLoweredName(prog, expr[b].c_str(), &low) == HIPRTC_SUCCESS && low;
            if (ok) names[b] = low;
        }
        size_t cs = 0;
        if (ok && hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs) {
            code.resize(cs);
            if (hiprtcGetCode(prog, code.data()) != HIPRTC_SUCCESS) code.clear();
        }
    }
    (void)hiprtcDestroyProgram(&prog);
    return code;
}

// Loads (compiling on first use per process) the specialisation for this net on the current
// device.  One module per (device, shape, act, outk), kept for the life of the process.
inline const Specialised *get_middle4(int device, const int *dims, int L, int act, int outk, bool with_slabs, bool bf16, size_t lds_bytes) {
    if (bf16 && !with_slabs) return nullptr;
    static std::mutex mu;
    static std::map<std::string, Specialised> cache;
    const std::string key = std::to_string(device) + "|" + shape_list(dims, L) + "|" + std::to_string(act) + "|" + std::to_string(outk) +
                            (with_slabs ? "|s" : "") + (bf16 ? "|bf16" : "");
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return (it->second.fn[1] || it->second.fn[2]) ? &it->second : nullptr;
    Specialised &sp = cache[key];
    std::string names[3];
    sp.n_fn = with_slabs ? 3 : 2;
    const std::vector<char> code = compile_middle4(dims, L, act, outk, with_slabs, bf16, names, &sp.log);
    if (code.empty()) return nullptr;
    if (hipModuleLoadData(&sp.module, code.data()) != hipSuccess) { (void)hipGetLastError(); sp.log += "\nhipModuleLoadData failed"; return nullptr; }
    for (int b = (bf16 ? 2 : 0); b < sp.n_fn; b++) {
        if (hipModuleGetFunction(&sp.fn[b], sp.module, names[b].c_str()) != hipSuccess) {
            (void)hipGetLastError();
            sp.fn[0] = sp.fn[1] = sp.fn[2] = nullptr;
            sp.log += "\nhipModuleGetFunction failed";
            return nullptr;
        }
        // more than 64 KB of dynamic LDS needs the opt-in, exactly as for the ahead-of-time kernels
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sp.fn[b]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            if (lds_bytes > 64 * 1024) { // without the opt-in such a launch would be refused: keep the AOT kernels
                sp.fn[0] = sp.fn[1] = sp.fn[2] = nullptr;
                sp.log += "\nhipFuncSetAttribute(MaxDynamicSharedMemorySize) failed";
                return nullptr;
            }
        }
    }
    return &sp;
}

// rowblock_kernel<RbStaticShape<dims...>, act, outk> (rowblock_kernel.h): the two-launch step's training kernel, fn[0].
inline const Specialised *get_rowblock(int device, const int *dims, int L, int act, int outk, bool bf16, size_t lds_bytes) {
    static std::mutex mu;
    static std::map<std::string, Specialised> cache;
    const std::string key = std::to_string(device) + "|" + shape_list(dims, L) + "|" + std::to_string(act) + "|" + std::to_string(outk) + (bf16 ? "|bf16" : "");
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second.fn[0] ? &it->second : nullptr;
    Specialised &sp = cache[key];
    sp.n_fn = 1;
    const std::string expr = "gnn::rowblock_kernel<gnn::RbStaticShape<" + shape_list(dims, L) + ">, " + std::to_string(act) + ", " + std::to_string(outk) + ", false, 0, " + (bf16 ? "true" : "false") + ">";
    const std::string src = "#include \"rowblock_kernel.h\"\n";
    const char *hdr_src[] = {kEmbedded_kernels_h, kEmbedded_fused_kernels_h, kEmbedded_middle4_kernel_h, kEmbedded_rowblock_kernel_h};
    const char *hdr_name[] = {"kernels.h", "fused_kernels.h", "middle4_kernel.h", "rowblock_kernel.h"};
    hiprtcProgram prog = nullptr;
    if (hiprtcCreateProgram(&prog, src.c_str(), "gnn_rowblock_jit.hip", 4, hdr_src, hdr_name) != HIPRTC_SUCCESS) { sp.log = "hiprtcCreateProgram failed"; return nullptr; }
    (void)hiprtcAddNameExpression(prog, expr.c_str());
    const char *opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-kernarg-preload-count=16"}; // (as build.py: rowblock_kernel.h)
    const hiprtcResult rc = hiprtcCompileProgram(prog, 5, opts);
    size_t ls = 0;
    if (hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) { sp.log.assign(ls, '\0'); (void)hiprtcGetProgramLog(prog, &sp.log[0]); }
    std::vector<char> code;
    std::string name;
    if (rc == HIPRTC_SUCCESS) {
        const char *low = nullptr;
        size_t cs = 0;
        if (hiprtcGetLoweredName(prog, expr.c_str(), &low) == HIPRTC_SUCCESS && low) name = low;
        if (!name.empty() && hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs) {
            code.resize(cs);
            if (hiprtcGetCode(prog, code.data()) != HIPRTC_SUCCESS) code.clear();
        }
    }
    (void)hiprtcDestroyProgram(&prog);
    if (code.empty()) return nullptr;
    if (hipModuleLoadData(&sp.module, code.data()) != hipSuccess) { (void)hipGetLastError(); sp.log += "\nhipModuleLoadData failed"; return nullptr; }
    if (hipModuleGetFunction(&sp.fn[0], sp.module, name.c_str()) != hipSuccess) { (void)hipGetLastError(); sp.fn[0] = nullptr; return nullptr; }
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(sp.fn[0]), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (lds_bytes > 64 * 1024) { sp.fn[0] = nullptr; return nullptr; }
    }
    return &sp;
}

} // namespace jit
} // namespace gnn
