"""The slot schedule of k_pl_stem2xs (csrc/planes_stem2xs.hip: the fused plane stem as a row stream, lfd_resnet.py:376-413) restated
in Python and checked exhaustively: producer waves write stream pixels in groups of 32 into a ring of RR = 10 rows while consumer
waves read the five rows of the chunk produced one slot earlier -- with ONE workgroup barrier per slot this is only correct if
  (a) every pixel of the five rows a chunk reads was written in an EARLIER slot, and is still the newest thing in its ring row,
  (b) no ring row a consumer reads in a slot is written by a producer in the same slot,
for every way a workgroup's run of chunks can start inside a strip, cross strips (a new SEGMENT: 5 fresh rows) and end.
CPU test: pure arithmetic, the same formulas as the kernel (walk_next, the group ranges, the ring row of a pixel)."""
import itertools

import pytest

RR, IW, PROLOGUE_PX = 10, 33, 5 * 33


def walk(first_cy, nchunks, cy_per_strip):
    """(j = chunk of the segment, rc = ring row of the chunk's first row) per chunk of a run -- Walk / walk_next"""
    out, cy, j, rc = [], first_cy, 0, 0
    for _ in range(nchunks):
        out.append((j, rc))
        cy += 1
        if cy == cy_per_strip:
            cy, j, rc = 0, 0, rc + 5
        else:
            j, rc = j + 1, rc + 4
        if rc >= RR:
            rc -= RR
    return out


def produced_pixels(j):
    """stream pixels (segment row m, column mx) the producer slot of segment chunk j writes, and the junk-slot writes it skips"""
    if j == 0:
        q_first, ng, limit = 0, 6, PROLOGUE_PX
    else:
        ga, gb = (33 * (j - 1) + 7) >> 3, (33 * j + 7) >> 3
        q_first, ng, limit = PROLOGUE_PX + 32 * ga, gb - ga, 1 << 30
    assert 4 <= ng <= 6
    px = []
    for q in range(q_first, q_first + 32 * ng):
        if q < limit:
            px.append((q // IW, q % IW))
    return px


@pytest.mark.parametrize('cy_per_strip', [1, 2, 3, 7, 40, 135])
def test_ring_schedule_has_no_hazard_and_no_gap(cy_per_strip):
    for first_cy, nchunks in itertools.product(sorted({0, 1, cy_per_strip // 2, cy_per_strip - 1}), [1, 2, 3, 9, 41, 150]):
        run = walk(first_cy % cy_per_strip, nchunks, cy_per_strip)
        ring = {}                     # ring row -> {column: (segment id, segment row)} of the newest pixel written there
        seg, seg_of = -1, []
        for (j, rc) in run:
            seg += (j == 0)
            seg_of.append(seg)
        for slot in range(nchunks + 1):
            reads = set()
            if slot >= 1:             # consumers: chunk slot - 1, rows 4 j .. 4 j + 4 of its segment
                j, rc = run[slot - 1]
                for r in range(5):
                    rr = (rc + r) % RR
                    reads.add(rr)
                    row = ring.get(rr, {})
                    for mx in range(IW):
                        assert row.get(mx) == (seg_of[slot - 1], 4 * j + r), \
                            'slot %d: row %d column %d of chunk %d is not in the ring (found %s)' % (slot, 4 * j + r, mx, slot - 1, row.get(mx))
            if slot < nchunks:        # producers: chunk slot
                j, rc = run[slot]
                mlo = 4 * j + 1 if j else 0
                for (m, mx) in produced_pixels(j):
                    my = m - mlo
                    assert 0 <= my <= 4, 'the frame patch of a slot covers five mid rows'
                    rr = (rc + (1 if j else 0) + my) % RR
                    assert rr not in reads, 'slot %d: the producer writes ring row %d while the consumer reads it' % (slot, rr)
                    ring.setdefault(rr, {})[mx] = (seg_of[slot], m)


def test_group_counts_match_the_kernel_comment():
    """a segment's first chunk: 165 pixels in 6 groups; afterwards 132 new pixels per chunk in 4 groups, a fifth one every eighth chunk"""
    counts = [((33 * j + 7) >> 3) - ((33 * (j - 1) + 7) >> 3) for j in range(1, 200)]
    assert set(counts) == {4, 5} and counts.count(5) * 8 <= len(counts) + 8
    for j in range(1, 200):           # the rows a chunk needs are complete when its producer slot ends
        assert PROLOGUE_PX + 32 * ((33 * j + 7) >> 3) >= IW * (4 * j + 5)
