"""mi_bvh_build timings of the 0.9 M-triangle interior in the situations bench.py meets (MIW_DEBUG=1 prints the builder's own split):
first build of a process, repeated builds in one context, a fresh context after another scene's context was closed, and a fresh
context while another one holds a large sample log (the `extras` block of the default bench line)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from mitsuba2_amd import api, scenes
    torch.zeros(1, device="cuda")
    scene, sensor = scenes.interior_scene(1920, 1080, 16, device=-1)
    small, ssensor = scenes.cornell_box(1920, 1080, 16, diffuse_only=False, device=-1)

    def build(dev, what, desc):
        t0 = time.perf_counter(); dev.upload(desc); dt = (time.perf_counter() - t0) * 1e3
        c = dev.counters()
        print("%-58s upload + build %7.1f ms (wall), mi_bvh_build %7.1f ms (bvh4 %.2f, bvh8 %.2f)" % (what, dt, c.ms_bvh_build, c.ms_bvh4, c.ms_bvh8), flush=True)

    d = api.Device(0)
    build(d, "interior, first build of the process", scene.desc())
    build(d, "interior, same context again", scene.desc())
    build(d, "interior, same context, third time", scene.desc())
    d.close()
    d = api.Device(0); build(d, "interior, fresh context", scene.desc()); d.close()
    d = api.Device(0); build(d, "material balls, fresh context", small.desc())
    job = api.PathIntegrator().render_job(ssensor); film, st = d.render(job)          # allocates that scene's sample log
    d.close()
    d = api.Device(0); build(d, "interior, fresh context after the balls' was closed", scene.desc()); d.close()
    big = torch.empty(20 << 30, dtype=torch.uint8, device="cuda")                       # another owner of 20 GB, as the headline context's log
    d = api.Device(0); build(d, "interior, fresh context beside a 20 GB allocation", scene.desc()); d.close()
    del big; torch.cuda.empty_cache()
    d = api.Device(0); build(d, "interior, fresh context after freeing it", scene.desc()); d.close()
    # the `extras` order of bench.py: the balls at their 1 024 spp (a 34 GB sample log), that context closed, then the interior
    big_scene, big_sensor = scenes.cornell_box(1920, 1080, 1024, diffuse_only=False, device=-1)
    d = api.Device(0); build(d, "material balls, fresh context", big_scene.desc())
    job = api.PathIntegrator().render_job(big_sensor); cfg = job.cfg
    cfg.film_on_device = 1; cfg.film_f64 = 0; cfg.film_mode = 0; cfg.profile = 1; cfg.plan = 0; cfg.samples_per_launch = 1024
    film = torch.zeros(1080 * 1920 * 5, dtype=torch.float32, device="cuda")
    d.check(d.L.mi_render(d.ctx, C.byref(cfg), C.c_void_p(film.data_ptr()))); torch.cuda.synchronize()
    print("balls rendered at 1024 spp: log %.1f GB" % (d.counters().log_bytes / 2**30), flush=True)
    t0 = time.perf_counter(); d.close(); print("closing that context: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    d = api.Device(0); build(d, "interior, fresh context after the 34 GB log was freed", scene.desc())
    build(d, "interior, same context again", scene.desc()); d.close()


if __name__ == "__main__":
    main()
