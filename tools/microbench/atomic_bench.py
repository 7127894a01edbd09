import ctypes, os, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libatomic_bench.so"))
for n, nbins, label in [(2_700_000, 2268, "uniform random tiles"), (2_700_000, 8160, "uniform 8160 bins"), (2_700_000, 2268, "sorted-ish (spatially coherent)")]:
    idx = torch.randint(0, nbins, (n,), device="cuda", dtype=torch.int32)
    if "sorted" in label:
        idx = torch.sort(idx).values
    cnt = torch.zeros(nbins, device="cuda", dtype=torch.int32)
    out = torch.zeros(n, device="cuda", dtype=torch.int32)
    a, b = ctypes.c_float(), ctypes.c_float()
    torch.cuda.synchronize()
    rc = L.run(n, ctypes.c_void_p(idx.data_ptr()), ctypes.c_void_p(cnt.data_ptr()), ctypes.c_void_p(out.data_ptr()), nbins, 20, ctypes.byref(a), ctypes.byref(b))
    print(f"{label:36s} n={n} bins={nbins}: no-return {a.value*1e3:8.1f} us ({n/a.value/1e6:7.1f} G/s)   returning {b.value*1e3:8.1f} us ({n/b.value/1e6:7.1f} G/s)  rc={rc}")
