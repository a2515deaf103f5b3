import os, sys, subprocess, json
root=os.getcwd()
for lib in ("libtangency.so","libtangency_diag2.so","libtangency.so","libtangency_diag2.so"):
    for cfg,W,steps in ((3,4096,4),(5,384,3)):
        env=dict(os.environ, TANGENCY_LIB=os.path.join(root,"incorporating_different_sources_amd",lib))
        r=subprocess.run([sys.executable,"bench.py","--config",str(cfg),"--windows",str(W),"--steps",str(steps),"--warmup","1","--no-cpu-baseline","--no-end-to-end","--no-general-layout"],env=env,capture_output=True,text=True)
        d=json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        print(lib, "config",cfg, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["parity_max_abs_diff_vs_oracle"], flush=True)
