import ctypes, os, sys, faulthandler
faulthandler.enable()
import torch
torch.zeros(1, device="cuda")
hip = ctypes.CDLL("libamdhip64.so.7")       # the one torch already loaded (same SONAME)
print("runtime:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1], flush=True)
class Loc(ctypes.Structure): _fields_ = [("type", ctypes.c_int), ("id", ctypes.c_int)]
class Prop(ctypes.Structure): _fields_ = [("type", ctypes.c_int), ("requestedHandleType", ctypes.c_int), ("location", Loc), ("win32", ctypes.c_void_p), ("compressionType", ctypes.c_ubyte), ("gpuDirect", ctypes.c_ubyte), ("usage", ctypes.c_ushort)]
prop = Prop(); prop.type = 1; prop.requestedHandleType = 1; prop.location.type = 1; prop.location.id = 0
g = 64 << 20
h = ctypes.c_void_p()
print("create", hip.hipMemCreate(ctypes.byref(h), ctypes.c_size_t(g), ctypes.byref(prop), ctypes.c_ulonglong(0)), flush=True)
fd = ctypes.c_int(-1)
print("export", hip.hipMemExportToShareableHandle(ctypes.byref(fd), h, 1, ctypes.c_ulonglong(0)), fd.value, flush=True)
h2 = ctypes.c_void_p()
print("import ...", flush=True)
v = ctypes.c_int(0); hip.hipRuntimeGetVersion(ctypes.byref(v)); print("runtime version", v.value, flush=True)
print("import", hip.hipMemImportFromShareableHandle(ctypes.byref(h2), ctypes.byref(fd), 1), flush=True)
p = ctypes.c_void_p()
print("reserve", hip.hipMemAddressReserve(ctypes.byref(p), ctypes.c_size_t(g), ctypes.c_size_t(0), None, ctypes.c_ulonglong(0)), flush=True)
print("map", hip.hipMemMap(p, ctypes.c_size_t(g), ctypes.c_size_t(0), h2, ctypes.c_ulonglong(0)), flush=True)
class Acc(ctypes.Structure): _fields_ = [("location", Loc), ("flags", ctypes.c_int)]
acc = Acc(); acc.location.type = 1; acc.location.id = 0; acc.flags = 3
print("access", hip.hipMemSetAccess(p, ctypes.c_size_t(g), ctypes.byref(acc), ctypes.c_size_t(1)), flush=True)
class Mem:
    __cuda_array_interface__ = {"shape": (1024,), "typestr": "<i4", "data": (p.value, False), "version": 2}
t = torch.as_tensor(Mem(), device="cuda")
t.fill_(7)
print("tensor over the imported chunk:", t[:4].cpu().tolist(), flush=True)
