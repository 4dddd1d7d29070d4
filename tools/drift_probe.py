import importlib, os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mvr = importlib.import_module("multi-view-registration_amd")
V, n = 12, 200000
sp = mvr.synth_params(V, 3); piv, ax = mvr.synth_prior(sp); origin = np.array(sp.pivot)
with mvr.Context(0) as ctx:
    for v in range(V): ctx.upload(V + v, mvr.synth_view(sp, v, n))
    poses = [np.eye(4)] + [mvr.axis_rotation(piv, ax, mvr.turntable_angle(v, V)) for v in range(1, V)]
    edges = [(v, (v + 1) % V) for v in range(V)]
    done = 0
    for chunk in (30, 170, 800, 3000):
        ctx.sync(); t0 = time.perf_counter()
        poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 4.0, origin, steps=chunk)
        ctx.sync(); dt = time.perf_counter() - t0
        done += chunk
        print("steps %d..%d: %.4f ms/step, n_corr %d, mse %.6g" % (done - chunk, done, 1e3 * dt / chunk, sum(info["pair_n"]), np.mean(info["pair_mse"])), flush=True)
        ctx.tune(grid_debug=1)
        poses, info = ctx.ring_step(list(range(V)), [V + v for v in range(V)], edges, poses, 4.0, origin, steps=1)
        ctx.tune(grid_debug=0)
        done += 1
