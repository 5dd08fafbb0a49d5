import os, sys, time
sys.path.insert(0, ".")
from cellularautomatons3d_amd import Engine, host
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
e = Engine(0); e.set_option("stats", 0)
e.configure(G); e.set_rule_strings()
e.upload_state(host.random_fill(host.words_per_buffer(G)))
steps = 32 if G >= 2048 else 256
e.step(steps); e.synchronize()
t0 = time.perf_counter(); e.step(steps); e.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"G {G}: {dt*1e6:.2f} us/step frac {0.25*G**3/dt/8e12:.3f} {e.info().kernel_name.decode()}")
