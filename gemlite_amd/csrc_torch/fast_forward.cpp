// fast_forward.cpp — the eager host path of GemLiteLinear.forward in C++.
//
// The reference launches a Triton kernel from Python (gemlite/core.py:128-195: forward_functional -> GEMLITE_TRITON_MAPPING[...].forward);
// this package's Python path builds a gemlite_hip_forward_args per call through ctypes, 14 us per `layer(x)` for a 4.7 us kernel
// (bench line `eager`).  This extension keeps, per layer, the immutable byte image of the struct with the static fields filled in
// (the very template core._build_template() makes) and does the per-call part without Python between tensor and launch: validate
// that the layer's tensors are still the ones the template was built from, allocate the output, fill the five per-call fields,
// fetch torch's current HIP stream and the per-(device, stream) workspace, ONE call of gemlite_hip_forward.  Anything unusual —
// changed tensors, a table / tuning epoch change, non-contiguous or wrong-dtype x, an unknown M for a loaded tuning table, any
// error status — returns None and the Python path (which raises the reference's exceptions) runs instead.
// CPython C API + ATen only (no pybind11 dispatch in the call path).  Built in-tree by gemlite_amd/csrc_torch/build.py.
#include <Python.h>

#include <ATen/ATen.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/python_variable.h>

#include <array>
#include <atomic>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/gemlite_hip.h"

namespace {

struct FastLayer {
    gemlite_hip_forward_args tmpl;
    const void *wq, *sc, *zr;  // data pointers the template was built from
    int dev;
    c10::ScalarType x_dtype, out_dtype;
    int64_t epoch;
    // tuning per M: the loaded table / autotune result at `epoch` (absent key: ask Python once; has == false: library default)
    struct Tune { bool has; std::array<int32_t, 4> t; };
    std::unordered_map<int64_t, Tune> tuning;
    bool table_empty;  // no tuning table loaded at `epoch`: never ask
    std::mutex mu;     // guards `tuning` (two threads may run the same layer)
};

const char* CAPSULE = "gemlite_amd.FastLayer";

void capsule_free(PyObject* c) { delete (FastLayer*)PyCapsule_GetPointer(c, CAPSULE); }

// ---- per-(device, stream) workspace, zero-filled once (the kernels leave the counters zero) ----------------------------------------
struct WsKey {
    int dev;
    void* stream;
    bool operator<(const WsKey& o) const { return dev != o.dev ? dev < o.dev : stream < o.stream; }
};
std::mutex g_ws_mutex;
std::map<WsKey, at::Tensor> g_ws;
// Replaced (outgrown) workspaces are RETIRED, never released: another thread may still hold the raw pointer of the old
// buffer in its thread-local view below and launch with it — split-K tickets and slabs written into memory the caching allocator had
// already handed to an activation tensor would corrupt it silently.  A process grows a workspace a handful of times (geometrically),
// so the retired list stays at a few MB.  g_ws_generation tells the thread-local views to look again.
std::vector<at::Tensor> g_ws_retired;
std::atomic<uint64_t> g_ws_generation{1};

at::Tensor workspace_for(int dev, void* stream, uint64_t nbytes) {
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(WsKey{dev, stream});
    if (it != g_ws.end() && (uint64_t)it->second.numel() >= nbytes) return it->second;
    uint64_t size = nbytes < (1ull << 20) ? (1ull << 20) : nbytes;
    uint64_t p2 = 1;
    while (p2 < size) p2 <<= 1;  // grow geometrically
    at::Tensor ws = at::zeros({(int64_t)p2}, at::TensorOptions().dtype(at::kByte).device(at::kCUDA, dev));
    if (it != g_ws.end()) g_ws_retired.push_back(it->second);
    g_ws[WsKey{dev, stream}] = ws;
    g_ws_generation.fetch_add(1, std::memory_order_release);
    return ws;
}

c10::ScalarType scalar_type_of_code(int code, bool* ok) {  // gemlite dtype codes (include/gemlite_hip.h) -> ATen
    *ok = true;
    switch (code) {
        case GEMLITE_DT_FP32: return at::kFloat;
        case GEMLITE_DT_FP16: return at::kHalf;
        case GEMLITE_DT_BF16: return at::kBFloat16;
        default: *ok = false; return at::kFloat;
    }
}

// make(template_bytes, W_q, scales, zeros, epoch, table_empty) -> capsule
PyObject* fast_make(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 6 || !PyBytes_Check(args[0]) || !THPVariable_Check(args[1]) || !THPVariable_Check(args[2]) || !THPVariable_Check(args[3])) {
        PyErr_SetString(PyExc_TypeError, "make(template: bytes, W_q, scales, zeros, epoch: int, table_empty: bool)");
        return nullptr;
    }
    if ((size_t)PyBytes_GET_SIZE(args[0]) != sizeof(gemlite_hip_forward_args)) {
        PyErr_SetString(PyExc_ValueError, "template size != sizeof(gemlite_hip_forward_args): header / library mismatch");
        return nullptr;
    }
    auto* L = new FastLayer();
    std::memcpy(&L->tmpl, PyBytes_AS_STRING(args[0]), sizeof(L->tmpl));
    const at::Tensor& wq = THPVariable_Unpack(args[1]);
    const at::Tensor& sc = THPVariable_Unpack(args[2]);
    const at::Tensor& zr = THPVariable_Unpack(args[3]);
    L->wq = wq.data_ptr();
    L->sc = sc.numel() ? sc.data_ptr() : nullptr;
    L->zr = zr.numel() ? zr.data_ptr() : nullptr;
    L->dev = wq.is_cuda() ? (int)wq.device().index() : -1;
    L->epoch = PyLong_AsLongLong(args[4]);
    L->table_empty = PyObject_IsTrue(args[5]) == 1;
    bool ok_in = false, ok_out = false;
    L->x_dtype = scalar_type_of_code(L->tmpl.input_dtype, &ok_in);
    L->out_dtype = scalar_type_of_code(L->tmpl.output_dtype, &ok_out);
    if (L->dev < 0 || !ok_in || !ok_out || L->tmpl.struct_size != sizeof(gemlite_hip_forward_args)) {  // weight-only 16-bit layers only
        delete L;
        Py_RETURN_NONE;
    }
    return PyCapsule_New(L, CAPSULE, capsule_free);
}

// set_tuning(handle, M, None | (t0, t1, t2, t3))
PyObject* fast_set_tuning(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 3) { PyErr_SetString(PyExc_TypeError, "set_tuning(handle, M, tuning)"); return nullptr; }
    auto* L = (FastLayer*)PyCapsule_GetPointer(args[0], CAPSULE);
    if (!L) return nullptr;
    const int64_t M = PyLong_AsLongLong(args[1]);
    FastLayer::Tune tu{false, {0, 0, 0, 0}};
    if (args[2] != Py_None) {
        tu.has = true;
        for (int i = 0; i < 4; ++i) tu.t[i] = (int32_t)PyLong_AsLong(PyTuple_GetItem(args[2], i));
        if (PyErr_Occurred()) return nullptr;
    }
    {
        std::lock_guard<std::mutex> lock(L->mu);
        L->tuning[M] = tu;
    }
    Py_RETURN_NONE;
}

// forward(handle, W_q, scales, zeros, x, bias | None, matmul_type, epoch) -> Tensor | None (this call: take the Python path) | False (the
// handle is stale: rebuild it) | NotImplemented (the tuning of this M is not known yet: call set_tuning and retry)
PyObject* fast_forward(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 8) { PyErr_SetString(PyExc_TypeError, "forward(handle, W_q, scales, zeros, x, bias, matmul_type, epoch)"); return nullptr; }
    auto* L = (FastLayer*)PyCapsule_GetPointer(args[0], CAPSULE);
    if (!L) return nullptr;
    if (!THPVariable_Check(args[1]) || !THPVariable_Check(args[2]) || !THPVariable_Check(args[3]) || !THPVariable_Check(args[4])) Py_RETURN_NONE;
    const at::Tensor& wq = THPVariable_Unpack(args[1]);
    const at::Tensor& sc = THPVariable_Unpack(args[2]);
    const at::Tensor& zr = THPVariable_Unpack(args[3]);
    const at::Tensor& x = THPVariable_Unpack(args[4]);
    // stale handle (tuning table changed, tensors re-allocated in place by module.to() / param.data = ...): False = rebuild me
    if (PyLong_AsLongLong(args[7]) != L->epoch) Py_RETURN_FALSE;
    if (wq.data_ptr() != L->wq || (sc.numel() ? sc.data_ptr() : nullptr) != L->sc || (zr.numel() ? zr.data_ptr() : nullptr) != L->zr) Py_RETURN_FALSE;
    if (!x.is_cuda() || (int)x.device().index() != L->dev || x.scalar_type() != L->x_dtype || !x.is_contiguous() || x.dim() < 1 ||
        x.requires_grad())
        Py_RETURN_NONE;
    const int64_t K = x.size(-1);
    if (K != L->tmpl.K || K == 0) Py_RETURN_NONE;
    const int64_t M = x.numel() / K;
    if (M <= 0) Py_RETURN_NONE;
    const long mt = PyLong_AsLong(args[6]);
    gemlite_hip_forward_args a = L->tmpl;
    if (!L->table_empty) {
        std::lock_guard<std::mutex> lock(L->mu);
        auto it = L->tuning.find(M);
        if (it == L->tuning.end()) { Py_INCREF(Py_NotImplemented); return Py_NotImplemented; }
        if (it->second.has)
            for (int i = 0; i < 4; ++i) a.tuning[i] = it->second.t[i];
    }
    at::Tensor bias;
    if (args[5] != Py_None) {
        if (!THPVariable_Check(args[5])) Py_RETURN_NONE;
        bias = THPVariable_Unpack(args[5]);
    }
    c10::hip::OptionalHIPGuard guard;
    if (c10::hip::current_device() != L->dev) guard.set_index(L->dev);
    // output: x.shape[:-1] + (N,)
    c10::SmallVector<int64_t, 4> oshape(x.sizes().begin(), x.sizes().end());
    oshape.back() = a.N;
    at::Tensor out = at::empty(oshape, x.options().dtype(L->out_dtype));
    a.matmul_type = (int32_t)mt;
    a.x = x.data_ptr();
    a.out = out.data_ptr();
    a.M = M;
    a.stride_xm = K;
    a.stride_xk = 1;
    a.stride_om = a.N;
    a.stride_on = 1;
    void* stream = (void*)c10::hip::getCurrentHIPStream(L->dev).stream();
    int rc;
    {
        // thread-local view of the last workspace: no lock on the steady path.  It holds the TENSOR (a reference), not just its pointer,
        // and is refreshed whenever any workspace of the process was replaced (generation counter) — ADVICE r4
        thread_local int tl_dev = -1;
        thread_local void* tl_stream = nullptr;
        thread_local at::Tensor tl_ws;
        thread_local void* tl_ptr = nullptr;
        thread_local uint64_t tl_bytes = 0;
        thread_local uint64_t tl_gen = 0;
        const uint64_t gen = g_ws_generation.load(std::memory_order_acquire);
        if (tl_dev != L->dev || tl_stream != stream || !tl_ptr || tl_gen != gen) {
            tl_ws = workspace_for(L->dev, stream, 0);
            tl_gen = g_ws_generation.load(std::memory_order_acquire);
            tl_dev = L->dev; tl_stream = stream; tl_ptr = tl_ws.data_ptr(); tl_bytes = (uint64_t)tl_ws.numel();
        }
        a.workspace = tl_ptr;
        a.workspace_bytes = tl_bytes;
        rc = gemlite_hip_forward(&a, stream);
        if (rc == GEMLITE_ERR_WORKSPACE) {
            const uint64_t need = gemlite_hip_workspace_bytes(&a);
            tl_ws = workspace_for(L->dev, stream, need);
            tl_gen = g_ws_generation.load(std::memory_order_acquire);
            const at::Tensor& ws = tl_ws;
            tl_ptr = ws.data_ptr(); tl_bytes = (uint64_t)ws.numel();
            a.workspace = tl_ptr;
            a.workspace_bytes = tl_bytes;
            rc = gemlite_hip_forward(&a, stream);
        }
    }
    if (rc != GEMLITE_OK) Py_RETURN_NONE;  // the Python path repeats the call and raises the reference's exception class
    if (bias.defined()) out.add_(bias);
    return THPVariable_Wrap(std::move(out));
}

// workspace(device_index, stream_handle, nbytes) -> uint8 tensor (shared with the Python path: gemlite_amd._hip.workspace)
PyObject* fast_workspace(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    if (nargs != 3) { PyErr_SetString(PyExc_TypeError, "workspace(device_index, stream_handle, nbytes)"); return nullptr; }
    const int dev = (int)PyLong_AsLong(args[0]);
    void* stream = (void*)PyLong_AsUnsignedLongLong(args[1]);
    const uint64_t nbytes = PyLong_AsUnsignedLongLong(args[2]);
    if (PyErr_Occurred()) return nullptr;
    return THPVariable_Wrap(workspace_for(dev, stream, nbytes));
}

PyMethodDef methods[] = {
    {"make", (PyCFunction)(void (*)(void))fast_make, METH_FASTCALL, "per-layer launch template"},
    {"set_tuning", (PyCFunction)(void (*)(void))fast_set_tuning, METH_FASTCALL, "tuning[] of one M (or None)"},
    {"forward", (PyCFunction)(void (*)(void))fast_forward, METH_FASTCALL, "out = layer(x) without Python between tensor and launch"},
    {"workspace", (PyCFunction)(void (*)(void))fast_workspace, METH_FASTCALL, "per-(device, stream) split-K workspace"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_fast", "C++ eager host path of gemlite_amd (see fast_forward.cpp)", -1, methods};

}  // namespace

PyMODINIT_FUNC PyInit__fast(void) { return PyModule_Create(&module); }
