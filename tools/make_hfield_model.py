#!/usr/bin/env python3
"""Derive the BASELINE config-5 model (humanoid on height-field terrain) from the compiled benchmark model.

Replaces the floor plane (geom 0) of humanoid_mujoco_amd/assets/humanoid27.hbm by an 8x8 height field,
size "10 10 1 1" like the reference's terrain (simulation/assets/world.xml:14,58), with the deterministic
elevations of SURVEY.md §8(d) config 5: h[r,c] = 0.1 * Halton(1 + 8 r + c, 2) metres (bump <= 0.1 m =
MAX_FLOOR_BUMP_HEIGHT, simulation_parameters.py:48), shifted so the mean height is 0.
PGS runs exactly 50 sweeps (tolerance 0).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27.hbm")
DST = os.path.join(ROOT, "humanoid_mujoco_amd", "assets", "humanoid27_hfield.hbm")


def halton(i, b):
    f, r = 1.0 / b, 0.0
    while i > 0:
        r += f * (i % b)
        i //= b
        f /= b
    return r


def main():
    lines = open(SRC).read().splitlines()
    rec = {}
    order = []
    for ln in lines[1:]:
        if ln == "END":
            break
        k, name, rest = ln.split(" ", 2)
        rec[name] = [k, rest]
        order.append(name)

    def setv(name, kind, values):
        if name not in rec:
            order.append(name)
        rec[name] = [kind, (("%d " % len(values)) if kind in "IDS" else "") + " ".join(values)]

    def arr(name):
        return rec[name][1].split()[1:]

    nrow = ncol = 8
    data = [0.1 * halton(1 + 8 * r + c, 2) for r in range(nrow) for c in range(ncol)]
    mean = sum(data) / len(data)
    setv("nhfield", "i", ["1"]); setv("nhfielddata", "i", [str(nrow * ncol)])
    setv("hfield_nrow", "I", [str(nrow)]); setv("hfield_ncol", "I", [str(ncol)]); setv("hfield_adr", "I", ["0"])
    setv("hfield_size", "D", ["10", "10", "1", "1"])
    setv("hfield_data", "D", ["%.17g" % v for v in data])  # elevation = data * size[2]
    t = arr("geom_type"); t[0] = "1"; setv("geom_type", "I", t)
    d = arr("geom_dataid"); d[0] = "0"; setv("geom_dataid", "I", d)
    p = arr("geom_pos"); p[2] = "%.17g" % (-mean); setv("geom_pos", "D", p)
    rb = arr("geom_rbound"); rb[0] = "%.17g" % ((10 ** 2 + 10 ** 2 + 1) ** 0.5); setv("geom_rbound", "D", rb)
    rec["tolerance"] = ["d", "0"]
    rec["iterations"] = ["i", "50"]
    with open(DST, "w") as f:
        f.write("HBM1\n")
        for name in order:
            f.write("%s %s %s\n" % (rec[name][0], name, rec[name][1]))
        f.write("END\n")
    print("wrote", DST, "mean elevation %.4f m" % mean)


if __name__ == "__main__":
    main()
