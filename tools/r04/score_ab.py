"""Score-pass variants side by side: default (16-lane groups x R = 10, one octet per wave), FADEHIP_SCORE_PERSIST=1
(persistent waves drawing octets by ticket), FADEHIP_SCORE_G8=1 (8-lane groups x R = 19, sixteen alignments per wavefront).
Each in a process of its own: solo launches (HIP-event time of the score pass, 1 slot, serial), the streamed two-slot rate,
and a hash of rs + alignments that must be the same for all three.   python tools/r04/score_ab.py [config] [reads per batch]"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def child(cfgname, n):
    import numpy as np
    import fade_amd
    import synthgen as sg
    from fade_amd import synth
    cfg = synth.config(cfgname)
    g = sg.Genome(cfg["n_contigs"], cfg["contig_len"], cfg["genome_seed"])
    ctx = fade_amd.Context(device=0)
    ctx.genome_upload(g.names, g.ascii_contigs())
    pinned = [ctx.pinned_batch(sg.with_bounds(sg.make_reads(g, n, 100 * (k + 1), cfg))) for k in range(4)]
    h = hashlib.sha256()
    solo = []
    for k in range(24):
        ctx.annotate_upload(0, pinned[k % 4])
        ctx.annotate_run(0, cfg["floor_len"], cfg["window"])
        rs, aln, st = ctx.annotate_results(0)
        if k < 4:
            a = np.sort(aln, order="read_idx")
            h.update(rs.tobytes() + a.tobytes() + np.asarray(st).tobytes())
        if k >= 4:
            solo.append(ctx.last_profile(0)["forward_ms"])
    # streamed, two slots
    reps = 60
    for phase in range(2):
        busy = [False, False]
        for slot in range(2):
            ctx.annotate_upload(slot, pinned[slot])
        t0 = time.perf_counter()
        for seq in range(reps):
            slot = seq % 2
            if busy[slot]:
                ctx.annotate_results(slot)
            ctx.annotate_run(slot, cfg["floor_len"], cfg["window"])
            busy[slot] = True
            if seq + 2 < reps:
                ctx.annotate_upload(slot, pinned[(seq + 2) % 4])
        for slot in range(2):
            if busy[slot]:
                ctx.annotate_results(slot)
        dt = time.perf_counter() - t0
    print(json.dumps(dict(score_ms_solo_mean=float(np.mean(solo)), score_ms_solo_min=float(np.min(solo)), streamed_reads_per_s=n * reps / dt,
                          sha=h.hexdigest())))
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    res = {}
    for name, env in (("default", {}), ("persistent", {"FADEHIP_SCORE_PERSIST": "1"}), ("g8_r19", {"FADEHIP_SCORE_G8": "1"}), ("g8_r19_2waves", {"FADEHIP_SCORE_G8": "2"}),
                      ("default_again", {})):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", cfgname, str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, **env), timeout=600)
        if p.returncode != 0:
            res[name] = dict(error=p.stderr.decode()[-600:])
        else:
            res[name] = json.loads(p.stdout.decode().strip().splitlines()[-1])
        print(name, res[name], flush=True)
    shas = {v.get("sha") for v in res.values() if "sha" in v}
    res["same_results"] = len(shas) == 1
    print("same results:", res["same_results"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "score_ab_%s.json" % cfgname), "w"), indent=1)
