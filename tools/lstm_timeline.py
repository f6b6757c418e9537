"""Segment timeline of the split LSTM kernel's wavefronts 0 and 4 (one SIMD) of workgroup 0, from a PROBE build (-DTACO_LSTM_STAMPS [-DTACO_LSTM_SYNC=n]):
    TACO_ENV_LIB=build/ab/lib_probe.so python tools/lstm_timeline.py
Per timestep of blocks 0..: [start of chains A (behind the wait), end of chains A, end of cells A, start of chains B (behind x chain + wait), end of chains B,
end of cells B, behind the barrier]; printed as offsets in shader clocks from wavefront 0's first stamp, the two wavefronts side by side."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from taco_amd import policy as P, _lib
import test_policy_gpu as TP
lib = _lib.load()
st = torch.zeros(256, dtype=torch.int64, device="cuda")
lib.taco_debug_set_lstm_stamps.argtypes = [C.c_void_p]
rng = np.random.default_rng(0)
pol = P.ActorCritic(TP._random_policy(rng, 1, 5, [128, 128, 128], 128, [128, 128]), 1, 5)
fr = torch.randn(37, 4096, 26, device="cuda")
for _ in range(5):
    pol.values_ring(fr)
lib.taco_debug_set_lstm_stamps(st.data_ptr())
pol.values_ring(fr)
torch.cuda.synchronize()
lib.taco_debug_set_lstm_stamps(None)
s = st.cpu().numpy().reshape(2, 128)
t0 = s[0, 0]
names = ["M_A start", "M_A end", "C_A end", "M_B start", "M_B end", "C_B end", "barrier"]
print("stamp            wave 0   (d)      wave 4   (d)")
for i in range(0, 70):
    a, b = s[0, i] - t0, s[1, i] - t0
    da = s[0, i] - s[0, i - 1] if i else 0
    db = s[1, i] - s[1, i - 1] if i else 0
    if s[0, i] == 0:
        break
    print(f"{i // 7:2d} {names[i % 7]:10s} {a:8d} {da:6d}   {b:8d} {db:6d}")
