import os, sys, subprocess
for lib in [None, "tools/probe/fd2.bin", "tools/probe/fd3.bin", "tools/probe/fdd2.bin", "tools/probe/fdd3.bin"]:
    env = dict(os.environ)
    if lib: env["BG_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "tools/dyn_roofline.py"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    import json
    print(lib, [ (json.loads(l)["num_envs"], round(json.loads(l)["avg_launch_us"],1), round(json.loads(l)["frac_of_8TBps"]*100,1)) for l in lines] if lines else r.stderr[-300:])
