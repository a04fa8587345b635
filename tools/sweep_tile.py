"""Time of one LAIR iteration (C ABI, MiniBooNE shape) against the batch size for the row tiles of the split-bf16
MLP kernel.  One process per setting (the choice is read from the environment at first use):

    python tools/sweep_tile.py            # runs itself once per (CSVAE_SPLIT_TILE, CSVAE_SPLIT16_WAVES)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BATCHES = (512, 1024, 2048, 4096, 8192, 16384, 32768, 65536)


def child():
    import torch
    import bench
    dev = torch.device("cuda", 0)
    out = []
    for B in BATCHES:
        b = bench.LairCabi("miniboone", B, 0, dev)
        steps = 40 if B <= 8192 else 15
        el, evs = b.timed(steps, 3)
        ms = lambda i, j: sum(e[i].elapsed_time(e[j]) for e in evs) / len(evs)
        out.append(f"{B}:{el / steps * 1e3:.4f}/{ms(0, 1):.4f}/{ms(1, 2):.4f}")
        del b
        torch.cuda.empty_cache()
    print(os.environ.get("CSVAE_SPLIT_TILE"), os.environ.get("CSVAE_SPLIT16_WAVES", "-"), " ".join(out))


if __name__ == "__main__":
    if os.environ.get("CS_SWEEP_CHILD"):
        child()
    else:
        print("tile waves  B:step_ms/enc_ms/dec_ms ...")
        for tile, waves in (("32", None), ("16", "4"), ("16", "8")):
            env = dict(os.environ, CS_SWEEP_CHILD="1", CSVAE_SPLIT_TILE=tile)
            if waves:
                env["CSVAE_SPLIT16_WAVES"] = waves
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=True)
