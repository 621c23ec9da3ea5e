"""Discrete simulation of one wavefront of k_path_phased (csrc/device/phased_kernel.h) under its vote policy.

    python tools/phase_sim.py [--tree bvh2|bvh4] [--segments 4000] [--seed 1]

What it is for: choosing the policy constants without burning GPU time, and backing the numbers DESIGN.md section 4 quotes
(max-count voting against threshold policies; where the shade vote's optimum sits once the walks get cheaper). It is a MODEL:
a body run costs a fixed number of wave cycles whatever its lane count (what the phase clock of a -DMIW_PHASE_STATS=1 build
measured on the material balls: node step 1 124, triangle test 2 319, shade 27 980 cycles per run over the BVH2), every lane
repeats  shade -> E walk -> [S walk] -> shade, and a walk is a random sequence of node steps and leaf visits whose means are
the measured ones (13.5 node steps + 3.2 triangle tests per closest-hit ray, 7.6 + 0.9 per shadow ray) with a log-normal tail
(sigma 1.0: the slowest lane of a wave takes ~10 x the mean, as measured). `--tree bvh4` walks 0.55 x the node steps at 1.35 x
the cycles per step and tests two triangles of a leaf per trip (the round-2 kernel).

Printed: lane-segments per million wave cycles for
    lock-step          max over lanes of (walk E) + max (walk S) + shade per iteration — k_path_resident
    phase machine      the kernel's vote, for several shade-vote ratios num : den (shade once n_shade * num >= den * lead)
    ideal              all lanes always busy: sum of work / 64
    pooled shading     the next step DESIGN.md section 9 costs: the wavefronts of a workgroup pool their shade-ready lanes in an LDS queue
                       and shade 64 at a time (see pooled()); per wavefront, so comparable with the rows above
"""
import argparse

import numpy as np

NODE, TRI, SHADE = 0, 1, 2


class Costs:
    def __init__(self, tree):
        self.shade = 27980.0
        if tree == "bvh2":
            self.node, self.tri, self.steps, self.pair = 1124.0, 2319.0, 1.0, False
        else:
            self.node, self.tri, self.steps, self.pair = 1124.0 * 1.35, 2319.0 * 1.1, 0.55, True


def draw_walk(rng, mean_nodes, mean_tris, steps_scale):
    """-> list of ops: NODE entries and ('leaf', n_tris) visits in walk order."""
    n = max(1, int(round(rng.lognormal(np.log(mean_nodes * steps_scale) - 0.5, 1.0))))
    leaves = rng.poisson(mean_tris / 2.2)                       # ~2.2 triangles per visited leaf (SAH, max 4)
    ops = [NODE] * n
    for _ in range(leaves):
        ops.insert(rng.integers(1, len(ops) + 1), ("leaf", int(rng.integers(1, 5))))
    return ops


class Lane:
    def __init__(self, rng, costs, p_shadow):
        self.rng, self.c, self.p_shadow = rng, costs, p_shadow
        self.state, self.ops, self.tris, self.walks_left = "shade", [], 0, 0
        self.done_segments = 0

    def begin(self):
        """after a shade run: queue the E walk and maybe the S walk"""
        self.walks = [draw_walk(self.rng, 13.5, 3.2, self.c.steps)]
        if self.rng.random() < self.p_shadow:
            self.walks.append(draw_walk(self.rng, 7.6, 0.9, self.c.steps))
        self.ops = self.walks.pop(0)
        self.state, self.tris = "walk", 0

    def ready(self):
        """which body this lane can run next: NODE, TRI, SHADE"""
        if self.state == "shade":
            return SHADE
        if self.tris > 0 and (not self.ops or self.ops[0] != NODE):
            return TRI                                           # holds a range and cannot descend further (Spec: may if next is a node)
        if self.ops and self.ops[0] == NODE:
            return NODE
        if self.tris > 0:
            return TRI
        return None

    def can_tri(self):
        return self.state == "walk" and self.tris > 0

    def step(self, body):
        if body == SHADE:
            self.done_segments += 1
            self.begin()
        elif body == NODE:
            self.ops.pop(0)
            if self.ops and self.ops[0] != NODE and self.tris == 0:     # a leaf becomes the lane's range
                self.tris = self.ops.pop(0)[1]
        elif body == TRI:
            self.tris -= 2 if (self.c.pair and self.tris >= 2) else 1
            if self.tris == 0 and self.ops and self.ops[0] != NODE:     # the next leaf was waiting
                self.tris = self.ops.pop(0)[1]
        if self.state == "walk" and not self.ops and self.tris == 0:
            if self.walks:
                self.ops = self.walks.pop(0)                            # E -> S turn (free in the model)
            else:
                self.state = "shade"


def phase_machine(rng, costs, segments, num, den, p_shadow=0.85):
    lanes = [Lane(rng, costs, p_shadow) for _ in range(64)]
    t = 0.0
    total = 0
    while total < segments:
        n_node = sum(1 for l in lanes if l.state == "walk" and l.ops and l.ops[0] == NODE)
        n_tri = sum(1 for l in lanes if l.can_tri())
        n_shade = sum(1 for l in lanes if l.state == "shade")
        lead = max(n_node, n_tri)
        if n_shade > 0 and n_shade * num >= lead * den:
            body, cost = SHADE, costs.shade
        elif n_node >= n_tri and n_node > 0:
            body, cost = NODE, costs.node
        else:
            body, cost = TRI, costs.tri
        for l in lanes:
            if body == SHADE and l.state == "shade":
                l.step(SHADE)
            elif body == NODE and l.state == "walk" and l.ops and l.ops[0] == NODE:
                l.step(NODE)
            elif body == TRI and l.can_tri():
                l.step(TRI)
        t += cost
        total = sum(l.done_segments for l in lanes)
    return total / t * 1e6


def pooled(rng, costs, segments, waves=4, xfer=300.0, p_shadow=0.85):
    """The step DESIGN.md section 9 costs: the `waves` wavefronts of a workgroup pool their shade-ready lanes. A lane whose walks
    are over DEPOSITS its path state in an LDS queue (one body run of `xfer` cycles for all such lanes of the wave: ~37 dwords
    each, wave-wide LDS stores) and becomes empty; empty lanes are REFILLED from the queue of shaded, walk-ready states (`xfer`
    again); a wavefront that finds 64 deposited states — or whatever is there when it has nothing else to run — runs ONE shade
    body over them (the usual shade cost, whatever the count) and puts the results on the walk-ready queue. Every wavefront has
    its own clock (they sit on different SIMDs); the pools are shared and free. Returns lane-segments per million cycles of the
    slowest wavefront, to be compared with `waves` independent phase machines (the same number: they do not interact)."""
    W = [[Lane(rng, costs, p_shadow) for _ in range(64)] for _ in range(waves)]
    for w in W:
        for l in w:
            l.begin()                                            # start with every lane walking
    clock = [0.0] * waves
    to_shade, ready = 0, 0                                       # states waiting for a shade run / shaded states waiting for a lane
    done = 0
    while done < segments:
        i = min(range(waves), key=lambda k: clock[k])
        w = W[i]
        n_node = sum(1 for l in w if l.state == "walk" and l.ops and l.ops[0] == NODE)
        n_tri = sum(1 for l in w if l.can_tri())
        n_dep = sum(1 for l in w if l.state == "shade")          # walks over: state to deposit
        n_empty = sum(1 for l in w if l.state == "empty")
        lead = max(n_node, n_tri)
        if to_shade >= 64 or (to_shade > 0 and lead == 0 and n_dep == 0 and ready == 0):
            k = min(64, to_shade)                                # a full (or the last) shade run out of the pool
            to_shade -= k; ready += k; done += k
            clock[i] += costs.shade + 2.0 * xfer
        elif n_dep > 0 and n_dep * 2 >= lead:                    # deposit once the finished lanes are half the busier walk group
            for l in w:
                if l.state == "shade":
                    l.state = "empty"
            to_shade += n_dep
            clock[i] += xfer
        elif n_empty > 0 and ready > 0 and (n_empty * 2 >= lead or lead == 0):
            k = min(n_empty, ready)
            ready -= k
            for l in w:
                if k and l.state == "empty":
                    l.begin(); k -= 1
            clock[i] += xfer
        elif n_node >= n_tri and n_node > 0:
            for l in w:
                if l.state == "walk" and l.ops and l.ops[0] == NODE:
                    l.step(NODE)
            clock[i] += costs.node
        elif n_tri > 0:
            for l in w:
                if l.can_tri():
                    l.step(TRI)
            clock[i] += costs.tri
        else:
            clock[i] += costs.node                               # nothing to do: wait a beat for the others
    return done / max(clock) * 1e6


def lock_step(rng, costs, segments, p_shadow=0.85):
    t, total = 0.0, 0

    def cost(ops):
        n = sum(1 for o in ops if o == NODE)
        k = sum(o[1] for o in ops if o != NODE)
        return n * costs.node + k * costs.tri
    while total < segments:
        e = max(cost(draw_walk(rng, 13.5, 3.2, 1.0)) for _ in range(64))
        s = max(cost(draw_walk(rng, 7.6, 0.9, 1.0)) if rng.random() < p_shadow else 0.0 for _ in range(64))
        t += e + s + costs.shade
        total += 64
    return total / t * 1e6


def ideal(rng, costs, segments, p_shadow=0.85):
    work = 0.0
    for _ in range(segments):
        for mean_n, mean_t, p in ((13.5, 3.2, 1.0), (7.6, 0.9, p_shadow)):
            if rng.random() < p:
                ops = draw_walk(rng, mean_n, mean_t, costs.steps)
                tris = sum(o[1] for o in ops if o != NODE)
                work += sum(1 for o in ops if o == NODE) * costs.node + (tris / 2 if costs.pair else tris) * costs.tri
        work += costs.shade
    return segments / (work / 64) * 1e6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tree", default="bvh4", choices=["bvh2", "bvh4"])
    ap.add_argument("--segments", type=int, default=4000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    c = Costs(a.tree)
    print("tree %s: node step %.0f, triangle trip %.0f, shade %.0f cycles per run" % (a.tree, c.node, c.tri, c.shade))
    print("lock-step (BVH2 walk)        %7.1f lane-segments / Mcycle" % lock_step(np.random.default_rng(a.seed), Costs("bvh2"), a.segments))
    for num, den in ((2, 1), (1, 1), (3, 4), (2, 3), (1, 2), (1, 3)):
        r = phase_machine(np.random.default_rng(a.seed), c, a.segments, num, den)
        print("phase machine, shade vote %d:%d  %7.1f   (shade once n_shade >= %.2f x the busier walk body)" % (den, num, r, den / num))
    print("ideal (all lanes busy)       %7.1f" % ideal(np.random.default_rng(a.seed), c, a.segments))
    for xfer in (150.0, 300.0, 600.0):
        r = pooled(np.random.default_rng(a.seed), c, a.segments * 4, waves=4, xfer=xfer)
        print("pooled shading, 4 waves, %3.0f cycles per state transfer run: %7.1f per wavefront  (to compare with the phase machine's rows)" % (xfer, r / 4))


if __name__ == "__main__":
    main()
