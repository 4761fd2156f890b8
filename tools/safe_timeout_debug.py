"""tests/test_gpu_deferred.py::test_a_paced_call_whose_wait_times_out_loses_no_block by hand: which deadlines pass, and which blocks of the
bus differ from the undisturbed run (run it with and without GROOVE_SAFE_STREAMS=1; docs/HISTORY.md section 10 item 15)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from groove_amd import entities as E, lib
import test_gpu_deferred as M
sel = np.arange(3000, dtype=np.int64)
outs = []
for disturbed in (False, True):
    ctx = E.Context(0)
    banks = M._mixed_banks(ctx, sel)
    bus = ctx.bus(6 * 256)
    log = []
    for b in range(6):
        if disturbed and b == 2:
            for k in range(16):
                try:
                    ctx.debug_spin(k, 700)
                except lib.GrooveError:
                    break
        for i, (inst, events) in enumerate(banks):
            if events.get(b) is not None:
                inst.handle_midi_events(events[b])
            if disturbed and b == 3:
                ctx.sync_timeout_ms = 100
            try:
                inst.render_mix_paced(bus, 256, accumulate=i > 0, at_frame=b * 256)
            except lib.GrooveError as e:
                log.append((b, i, str(e)[:90]))
            ctx.sync_timeout_ms = 20000
    outs.append(bus.download().copy())
    print("disturbed", disturbed, "timeouts", log, [inst.kernel_form(256, True)[:40] for inst, _ in banks])
    ctx.close()
d = outs[0] != outs[1]
for b in range(6):
    blk = d[b * 256:(b + 1) * 256]
    print("block", b, "differing samples", int(blk.sum()), "max abs diff", float(np.abs(outs[0][b*256:(b+1)*256] - outs[1][b*256:(b+1)*256]).max()))
