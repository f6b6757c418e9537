"""Row I (the rigid-body integrate that stands in for gym.simulate, vec_task_asymmetry.py:313) has no reference arithmetic to be pinned to:
PhysX is a missing binary.  What CAN be stated is the scheme's own error: oracle/row_i_ref.py flies the fp32 semi-implicit scheme and a
float64 RK4 solution of the spec's ODE side by side in closed loop (same rate PID, allocator, battery, rotor and aero models) and this
test asserts (a) first-order convergence in the sub-iteration count and (b) the size of the error at the default two sub-iterations.
The full 1 000-step table (profiles/r02_e_row_i_error.txt): 1.1e-3 m / 2.0e-3 rad hover-like and 5.6e-3 m / 3.1e-3 rad
for a flip-like start -- i.e. the integrator's truncation error, not fp32 round-off, is what separates this build from any other
integrator of the same spec, PhysX included; the 1e-5 parity bar of the north star is met against the ORACLE, which shares the scheme."""
from oracle import row_i_ref as R


def test_row_i_converges_at_first_order_and_its_error_is_known():
    steps = 150
    e = {s: R.closed_loop_error(s, steps=steps, n=8, seed=1, spin=10.0) for s in (1, 2, 4)}
    for s in (1, 2, 4):
        assert 2.0 < e[s]["final_height"] < 3.0, "the bodies are supposed to stay airborne"
    # first order in h = dt / sub-iterations: halving h halves the error (measured ratios 1.95 ... 2.1)
    for a, b in ((1, 2), (2, 4)):
        for k in ("pos_linf", "att_linf"):
            ratio = e[a][k] / e[b][k]
            assert 1.6 < ratio < 2.5, f"{k}: error ratio {ratio:.2f} between {a} and {b} sub-iterations is not first order"
    # size at the default (two sub-iterations), 150 steps from a 10 rad/s roll rate: ~4e-3 m, ~3e-3 rad
    assert 1e-4 < e[2]["pos_linf"] < 2e-2 and 1e-4 < e[2]["att_linf"] < 2e-2, (e[2]["pos_linf"], e[2]["att_linf"])
