#!/usr/bin/env python3
"""Can a distributed apply be captured in a hipGraph?  usage: dist_graph_capture_probe.py exchange|beginend|apply [capture mode]
ROCm 7.2 / MI355X, 1-GPU self-loop: `exchange` (pack, RCCL send/recv, unpack on ONE stream) captures, instantiates and replays;
`beginend` and `apply` (the exchange on the plan's side stream, forked from and joined to the capturing stream by events) crash
inside hipStreamEndCapture -- so the fused steps are not offered as graphs."""
import ctypes, sys, pathlib
import numpy as np
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
import bench
from gt4py_amd.distributed.halo import Decomposition
from gt4py_amd.distributed.native import NativeComm, NativeHaloExchanger
case = sys.argv[1]
comm = NativeComm(rank=0, world_size=1)
stream = torch.cuda.Stream()
dec = Decomposition((512, 64, 512), (1, 1), 0, halo=1, periodic=(False, True))
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
with torch.cuda.stream(stream):
    pairs = bench._device_fields(dec.local_shape, n_pairs=1, seed=1, origin=dec.origin)
    ex = NativeHaloExchanger(dec, np.float64, comm, single_phase=False).tune("join", 0)
    inp, out = pairs[0]
    bound = ex.make_dist_lap5(inp, out, dec.origin, dec.origin)
    if case == "exchange":
        fn = lambda: ex.exchange(inp)
    elif case == "beginend":
        def fn():
            ex.begin(inp); ex.end()
    elif case == "apply":
        fn = bound
    fn(); fn()
    torch.cuda.synchronize()
    sp = ctypes.c_void_p(stream.cuda_stream)
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    print("begin", hip.hipStreamBeginCapture(sp, mode), flush=True)
    try:
        fn()
        print("enqueued", flush=True)
    except Exception as e:
        print("enqueue failed", repr(e)[:300], flush=True)
    g = ctypes.c_void_p()
    rc = hip.hipStreamEndCapture(sp, ctypes.byref(g))
    print("end", rc, hip.hipGetErrorString(rc), flush=True)
    if rc == 0:
        ge = ctypes.c_void_p()
        print("inst", hip.hipGraphInstantiate(ctypes.byref(ge), g, None, None, 0), flush=True)
        for _ in range(3):
            print("launch", hip.hipGraphLaunch(ge, sp), flush=True)
        torch.cuda.synchronize()
        print("ok", flush=True)
