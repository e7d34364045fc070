#!/usr/bin/env python3
"""Extracts the only numbers the reference publishes for this path - the per-MPC-step solver times of the
N = 60 runs and the averages per horizon (3 problems x cold/warm x 6 horizons = 36 PANTR averages, and the
comparators') - from its plots into tests/golden/published_series.json (data, no reference text).

Run in the build container (needs /root/reference; the GPU box does not have it):
    python3 tests/golden/extract_published_series.py [/root/reference]

Source: images/mpc-<problem>-60-runtimes-mpc-last-cold-warm.pdf.svg (README.md:36,69,106), produced by
new-benchmarks-paper/perfplot-cold-warm.py:137 from the runs dodo.py:69-111 enumerates.  Procedure
(BASELINE.md section 1):
  * data = the dotted polylines (stroke-dasharray "10, 16.5"), 60 points each, colours in legend order
    (dodo.py:47-52): #1f77b4 PANTR, #ff7f0e PANOC+(50), #2ca02c Struc. PANOC+(50), #d62728 Ipopt; the left panel
    (x < 2000) is the cold run, the right one the warm run;
  * y calibration = the left axis' major ticks (the 35-unit strokes, drawn as clip rectangles 8 units thick), one
    decade apart.  Tick labels are outlined glyphs, so the decade of the LOWEST major tick is a constant below,
    read off a raster rendering once; the script checks it: the mean of the 60 points of every series must equal
    the N = 60 point of the companion plot of averages (images/...-avg-runtimes-quantiles-...) to 3 significant
    digits, for every solver - a wrong decade fails by a factor of ten.
The y coordinate of the PDF space grows with the value (no flip)."""
import json
import math
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "published_series.json")
SOLVERS = {"#1f77b4": "pantr", "#ff7f0e": "panoc", "#2ca02c": "strucpanoc", "#d62728": "ipopt"}
# log10 of the value at the lowest major tick of the left axis (read off a rendering; checked below)
LOWEST_DECADE = {("hanging_chain", "last"): -1, ("quadcopter", "last"): -1, ("realistic_quadcopter", "last"): -1,
                 ("hanging_chain", "avg"): -3, ("quadcopter", "avg"): -3, ("realistic_quadcopter", "avg"): -2}


def polyline(d):
    """Absolute points of an SVG path made of one 'm' followed by relative pairs."""
    tok = d.split()
    assert tok[0] == "m", tok[0]
    pts, x, y = [], 0.0, 0.0
    for t in tok[1:]:
        if "," not in t:  # a command letter inside the path: not a plain polyline
            return None
        dx, dy = map(float, t.split(","))
        x, y = x + dx, y + dy
        pts.append((x, y))
    return pts


def read_plot(problem, kind):
    name = {"last": "runtimes-mpc-last", "avg": "avg-runtimes-quantiles"}[kind]
    s = open(os.path.join(REF, "images", f"mpc-{problem}-60-{name}-cold-warm.pdf.svg")).read()
    clips = re.findall(r'<clipPath[^>]*>\s*<path[^>]*?d="m ([\d.]+),([\d.]+) h ([\d.]+) v ([\d.]+) h', s, flags=re.S)
    major = sorted({float(y) + float(h) / 2 for x, y, w, h in clips if float(w) == 35.0 and float(x) < 1000})
    per_decade = (major[-1] - major[0]) / (len(major) - 1)
    assert all(abs((b - a) - per_decade) < 0.02 for a, b in zip(major, major[1:])), major
    y0, dec0 = major[0], LOWEST_DECADE[(problem, kind)]
    out = {}
    for p in re.findall(r"<path\b[^>]*?>", s, flags=re.S):
        if "stroke-dasharray:10, 16.5" not in p:
            continue
        solver = SOLVERS[re.search(r"stroke:(#[0-9a-f]{6})", p).group(1)]
        pts = polyline(re.search(r'\bd="([^"]*)"', p, flags=re.S).group(1))
        if pts is None:  # (Ipopt's cold series of the hanging chain carries a pen-up for a failed step)
            continue
        start = "cold" if pts[0][0] < 2000 else "warm"
        out[(solver, start)] = [10.0 ** (dec0 + (y - y0) / per_decade) for x, y in pts]
    return out


def main():
    data = {"source": "images/mpc-<problem>-60-runtimes-mpc-last-cold-warm.pdf.svg of kul-optec/pantr-cdc2023-experiments "
                      "(README.md:36,69,106); extracted by tests/golden/extract_published_series.py",
            "unit": "seconds of solver time per MPC step (alpaqa stats.elapsed_time, mpc-driver.cpp:199-201,216-222); "
                    "unstated CPU, one thread", "horizon": 60, "steps": 60, "series": {},
            # the companion plots (README.md:33,66,103): average solver time per MPC step (60 steps) against the horizon
            "averages_source": "images/mpc-<problem>-60-avg-runtimes-quantiles-cold-warm.pdf.svg (README.md:33,66,103)",
            "average_horizons": [10, 20, 30, 40, 50, 60], "averages": {}}
    for problem in ("hanging_chain", "quadcopter", "realistic_quadcopter"):
        last, avg = read_plot(problem, "last"), read_plot(problem, "avg")
        data["series"][problem] = {}
        for (solver, start), v in sorted(last.items()):
            assert len(v) == 60, (problem, solver, start, len(v))
            # self-check of the calibration: the N = 60 point (the 6th) of the plot of averages
            if (solver, start) in avg:
                ref = avg[(solver, start)][5]
                mean = sum(v) / len(v)
                assert abs(mean / ref - 1) < 2e-3, (problem, solver, start, mean, ref)
            data["series"][problem].setdefault(solver, {})[start] = [float(f"{x:.4g}") for x in v]
        data["averages"][problem] = {}
        for (solver, start), v in sorted(avg.items()):  # six horizons per curve (dodo.py:70: 10, 20, ..., 60)
            assert len(v) == 6, (problem, solver, start, len(v))
            data["averages"][problem].setdefault(solver, {})[start] = [float(f"{x:.4g}") for x in v]
        print(problem, {k: sorted(v) for k, v in data["series"][problem].items()})
    with open(OUT, "w") as f:
        json.dump(data, f, indent=0)
    p = data["series"]["hanging_chain"]["pantr"]
    print("hanging_chain PANTR: cold mean %.4f s, warm mean %.4f s (BASELINE.md 1.1: 0.3454, 0.1301)"
          % (sum(p["cold"]) / 60, sum(p["warm"]) / 60))


if __name__ == "__main__":
    main()
