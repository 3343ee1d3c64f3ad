import time, json
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
h = hs[0]
for name, fn in (("power", lambda: amdsmi.amdsmi_get_power_info(h)),
                 ("clock", lambda: amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)),
                 ("metrics", lambda: {k: v for k, v in amdsmi.amdsmi_get_gpu_metrics_info(h).items() if any(s in k for s in ("gfxclk", "socket_power", "temperature_hotspot", "throttle", "uclk"))})):
    try:
        t = time.perf_counter(); r = fn(); dt = time.perf_counter() - t
        print(name, "%.2f ms" % (dt * 1e3), json.dumps(r, default=str)[:600])
    except Exception as e:
        print(name, "ERR", repr(e)[:200])
import torch
x = torch.rand(8192, 8192, device="cuda", dtype=torch.float64)
for i in range(30):
    y = x @ x
    if i % 5 == 0:
        torch.cuda.synchronize()
        try:
            print(json.dumps({k: v for k, v in amdsmi.amdsmi_get_gpu_metrics_info(h).items() if k in ("current_gfxclk", "current_gfxclks", "average_socket_power", "current_socket_power", "average_gfxclk_frequency")}, default=str)[:400])
            print(amdsmi.amdsmi_get_power_info(h), amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
        except Exception as e:
            print("ERR", e)
