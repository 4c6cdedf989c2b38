import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import sweep_parity as sp
import rustpotter_amd as ra
ctx = ra.BatchContext(0)
fam, seed, ci = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
# run a family for cases [ci, ci] only by monkeypatching range via seed trick: re-implement minimal drivers
if fam == "model":
    orig = sp.make_model_case
    sp.make_model_case = lambda rng: orig(np.random.default_rng([seed, 99, ci]))
    try: print(sp.run_model_sweep(ra, 1, seed, ctx=ctx))
    except AssertionError as e: print("FAIL", str(e)[:3000])
elif fam == "builder":
    import types
    src = open('/root/repo/tests/sweep_parity.py').read()
    # run builder sweep but only case ci: patch the rng seeding by wrapping default_rng
    real = np.random.default_rng
    np.random.default_rng = lambda a=None: real([seed, 88, ci]) if isinstance(a, list) and a[:2] == [seed, 88] else real(a)
    try: print(sp.run_builder_sweep(ra, ctx, 1, seed))
    except AssertionError as e: print("FAIL", str(e)[:3000])
elif fam == "train":
    real = np.random.default_rng
    np.random.default_rng = lambda a=None: real([seed, 99, 7, ci]) if isinstance(a, list) and a[:3] == [seed, 99, 7] else real(a)
    try: print(sp.run_train_sweep(ra, ctx, 1, seed, verbose=True))
    except AssertionError as e: print("FAIL", str(e)[:3000])
