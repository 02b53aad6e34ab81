import sys, json, subprocess, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import f3dgaus_amd as f
L = f._lib.lib()
import runpy
for opt in ([], [(b"debug_skip_all", 1)], [(b"render_queue", 0)]):
    for k, v in [(b"debug_skip_all", 0), (b"render_pretest", 1), (b"render_cull", 1), (b"render_queue", 1)] + opt:
        L.f3dg_set_option(k, v)
    sys.argv = ["bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"]
    import io, contextlib
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
    d = json.loads(buf.getvalue().strip().splitlines()[-1])
    print(opt, round(d["value"]), d["roofline"]["stage_ms_per_step"])
