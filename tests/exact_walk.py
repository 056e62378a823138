"""An INDEPENDENT, exact line walker (test infrastructure).

It shares no code and no floating point with oracle/ or the HIP path: the rule of the reference walk --
  "step along the axis with the smallest exit time; on a tie the HIGHER axis wins; an axis' exit time after k steps is
   initial + delta * k; an axis whose remaining key range is zero never steps"
  (/root/reference/ohm/LineWalkCompute.h:282-307, :345-413) --
is evaluated in exact integer arithmetic (cross-multiplied rationals) on lattice-aligned inputs: every coordinate is an
integer multiple of `unit = resolution / SUB` with a dyadic resolution, so the inputs are exact in fp64 too.

What "exit time" means exactly: with d = end - start, an axis' k-th exit is at ray parameter u_a(k) = (x_a + k * res) /
|d_a| (x_a = distance from the start point to its voxel's wall in the direction of travel).  The reference measures time
in metres, t = L * u with one positive L for all axes, so ORDER and TIES of the times are those of the u_a(k).

fp64 agrees with the exact order only if rounding cannot flip a comparison.  generate() therefore keeps a ray only if
every comparison the walk makes is either
  * a STRUCTURAL tie: the two axes have equal |d_a|, equal x_a and equal step counts -- then the reference's fp64
    arithmetic runs the very same operations on the very same magnitudes for both axes and the times are bit-equal; or
  * separated by a relative gap of more than 1e-9 (fp64 keeps ~1e-15 through the handful of operations involved).
A mathematically exact tie between axes of DIFFERENT slope (2:1 ...) is not reproducible in fp64 -- delta_b is not the
bit pattern of 2 * delta_a -- so the reference itself has no defined behaviour there; such rays are discarded (counted).
"""
import random

SUB = 64         # lattice units per voxel
REGION = 32      # voxels per region axis


def _floor_div(a, b):
    return a // b  # Python ints: floor division


def voxel_of(p_units):
    """Global voxel coordinate of a lattice point (map origin 0): plain floor."""
    return tuple(_floor_div(c, SUB) for c in p_units)


def key_of(voxel):
    """(region key, local key) of a global voxel coordinate, region (0,0,0) centred on the origin as in ohm
    (ohm/MapCoord.h:45-93: region = floor(p / R + 0.5)): global voxel v lies in region floor((v + 16) / 32)."""
    half = REGION // 2
    region = tuple(_floor_div(v + half, REGION) for v in voxel)
    local = tuple(v + half - REGION * r for v, r in zip(voxel, region))
    return region, local


class Undecidable(Exception):
    """The ray asks fp64 for a comparison it cannot be trusted with (see the module docstring)."""


def walk(start_units, end_units, rel_gap=1e-9):
    """Exact voxel sequence [(region, local), ...] of the segment start -> end, both given in lattice units, with the
    reference's default flags (start voxel and end voxel included).  Raises Undecidable for rays outside the contract."""
    s, e = tuple(start_units), tuple(end_units)
    d = tuple(b - a for a, b in zip(s, e))
    v0, v1 = voxel_of(s), voxel_of(e)
    remaining = [b - a for a, b in zip(v0, v1)]
    sign = [1 if c < 0 else 0 for c in d]                 # as the reference: dir < 0
    step_dir = [-2 * sg + 1 for sg in sign]
    # distance (lattice units) from the start point to the first wall in the direction of travel
    x = []
    for a in range(3):
        lo = v0[a] * SUB
        x.append((s[a] - lo) if sign[a] else (lo + SUB - s[a]))
    absd = [abs(c) for c in d]
    stepped = [0, 0, 0]

    def time(a):
        """u_a as (numerator, denominator), None == +infinity."""
        if remaining[a] == 0 or absd[a] == 0:
            return None
        return (x[a] + abs(stepped[a]) * SUB, absd[a])

    def less(a, b):
        """time[a] < time[b], exactly -- refusing what fp64 could get wrong."""
        ta, tb = time(a), time(b)
        if ta is None:
            return False            # inf < anything: false (inf < inf: false)
        if tb is None:
            return True
        lhs, rhs = ta[0] * tb[1], tb[0] * ta[1]
        if lhs == rhs:
            structural = absd[a] == absd[b] and x[a] == x[b] and abs(stepped[a]) == abs(stepped[b])
            if not structural:
                raise Undecidable("exact tie between axes of different slope / offset")
            return False
        if abs(lhs - rhs) <= rel_gap * max(lhs, rhs):
            raise Undecidable("near tie")
        return lhs < rhs

    def select():
        axis = 0
        axis = axis if less(axis, 1) else 1
        axis = axis if less(axis, 2) else 2
        return axis

    cur = list(v0)
    out = []
    limit = sum((1 << a) for a in range(3) if remaining[a] == 0)
    axis = select()
    guard = 0
    while limit < 7 and tuple(cur) != v1:
        out.append(key_of(tuple(cur)))
        cur[axis] += step_dir[axis]
        remaining[axis] -= step_dir[axis]
        stepped[axis] += step_dir[axis]
        if remaining[axis] == 0:
            limit |= 1 << axis
        axis = select()
        guard += 1
        if guard > 100000:
            raise AssertionError("exact walk does not terminate")
    out.append(key_of(v1))
    return out


def generate(count, seed=20260927, resolution=0.125):
    """Tie-rich lattice rays: [(start_xyz float, end_xyz float, expected keys)], and how many candidates were discarded
    as undecidable.  Families: axis-aligned, plane and space diagonals through centres / corners / generic offsets
    (structural ties), 2:1 and 3:1 slopes and random lattice segments at generic offsets, zero components, negative
    directions, origins near region boundaries so the walks cross regions."""
    rng = random.Random(seed)
    unit = resolution / SUB
    rays, discarded, tried = [], 0, 0
    offsets_sym = [0, SUB // 2, SUB // 4, 3 * SUB // 8]   # same offset on every axis: ties on diagonals
    slopes = [(1, 0, 0), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1), (2, 1, 0), (1, 2, 0), (2, 1, 1), (1, 1, 2),
              (3, 1, 0), (1, 3, 1), (3, 2, 1), (2, 0, 1), (0, 0, 1), (0, 1, 0), (2, 2, 1), (3, 3, 1), (1, 3, 3)]
    while len(rays) < count:
        tried += 1
        family = rng.randrange(4)
        # start voxel: anywhere within a few regions, biased towards region boundaries (|v + 16| % 32 near 0)
        base = [rng.choice([-49, -48, -17, -16, -15, -1, 0, 1, 15, 16, 17, 47, 48]) + rng.randrange(-3, 4) for _ in range(3)]
        if family == 0:      # structural ties: +-1 / 0 slopes, symmetric offset
            sl = rng.choice(slopes[:5] + slopes[13:15])
            sg = [rng.choice((-1, 1)) for _ in range(3)]
            n = rng.randrange(1, 70)
            off = rng.choice(offsets_sym)
            s = [b * SUB + off for b in base]
            e = [s[a] + sg[a] * sl[a] * n * SUB + (rng.choice((0, 0, SUB // 8)) if sl[a] else 0) * 0 for a in range(3)]
        elif family == 1:    # integer slopes at generic (non-tying) offsets
            sl = rng.choice(slopes)
            sg = [rng.choice((-1, 1)) for _ in range(3)]
            n = rng.randrange(1, 40)
            s = [b * SUB + rng.choice((5, 11, 19, 23, 37, 41, 53, 59)) for b in base]
            e = [s[a] + sg[a] * sl[a] * n * SUB for a in range(3)]
        elif family == 2:    # random lattice segments (generic), some components zero
            s = [b * SUB + rng.randrange(1, SUB) for b in base]
            e = [s[a] + (0 if rng.random() < 0.15 else rng.randrange(-40 * SUB, 40 * SUB + 1)) for a in range(3)]
        else:                # diagonals with per-axis different symmetric-looking offsets and lengths in half voxels
            sg = [rng.choice((-1, 0, 1)) for _ in range(3)]
            n = rng.randrange(1, 120)
            s = [b * SUB + rng.choice(offsets_sym) for b in base]
            e = [s[a] + sg[a] * n * (SUB // 2) for a in range(3)]
        if s == e:
            continue
        try:
            keys = walk(s, e)
        except Undecidable:
            discarded += 1
            continue
        rays.append((tuple(c * unit for c in s), tuple(c * unit for c in e), keys))
    return rays, discarded
