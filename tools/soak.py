import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "x3d-multigrid_amd"))
import torch
import train_x3d_kinetics_multigrid as tr
for dt in (torch.float32, torch.bfloat16):
    torch.cuda.reset_peak_memory_stats()
    t0 = time.time()
    steps, cps = tr.run(init_lr=0.0125, warmup_steps=50, max_epochs=12, batch_size=8, steps=0, max_steps_run=1200,
                        iterations_per_epoch=100, save_model="/tmp/soak_ck_", save_every=0, use_graph=True, log_every=200,
                        val_every=400, val_batches=1, val_batch_size=1, act_dtype=dt)
    torch.cuda.synchronize()
    print("SOAK", dt, "steps", steps, "clips/s %.1f" % cps, "wall %.1f s" % (time.time() - t0),
          "reserved %.2f GB peak %.2f GB" % (torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_reserved() / 2**30), flush=True)
