#!/usr/bin/env python3
"""List scheduling of a launch's tickets on the persistent workgroups: what finer work items at the END of a launch can buy.

    python tools/tail_sim.py [passes] [workgroup slots]        (default: 1580 512 = ld_triangle 10 000 x 5008 on 256 CUs)

Model: every workgroup draws the next ticket when it is done (the kernel's ticket counter); a whole pass takes 36 us +- 8 %
per workgroup +- 5 % per ticket, a half-height ticket 20 us (0.55: measured), a quarter 12 us (0.33: ASSUMED before it was
built; the kernel's quarters turned out dearer -- profiles/r05/quarter_tickets_sweep.log -- and are not in the product), a
sixteenth 4.5 us.  Tickets are handed out large to small.  Printed: the mean end of the launch over 20 seeds.  The point:
1580 passes are 3.09 rounds of 512, the fluid bound is 111 us, whole passes alone end at 136 us, the product's 128 halved
passes at 124 us -- and NO split of the tail gets below ~118 us, because the finer the items the more they cost, and the
workgroups that hold a whole pass when the whole passes run out still need their 36 us."""
import heapq
import random
import statistics
import sys

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1580
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
TW, TH, TQ, TF = 36.0, 20.0, 12.0, 4.5


def sim(tickets, seed):
    random.seed(seed)
    speed = [1 + random.uniform(-0.08, 0.08) for _ in range(W)]
    heap = [(0.0, i) for i in range(W)]
    end = 0.0
    for d in tickets:
        t, i = heapq.heappop(heap)
        t2 = t + d * speed[i] * (1 + random.uniform(-0.05, 0.05))
        end = max(end, t2)
        heapq.heappush(heap, (t2, i))
    return end


def mean_end(nh, nq, nf):
    tickets = [TW] * (n - nh - nq - nf) + [TH] * (2 * nh) + [TQ] * (4 * nq) + [TF] * (16 * nf)
    return statistics.mean(sim(tickets, s) for s in range(20))


print(f"{n} passes on {W} workgroups: fluid bound {n * TW / W:.1f} us")
for nh, nq, nf in ((0, 0, 0), (64, 0, 0), (128, 0, 0), (256, 0, 0), (128, 32, 0), (128, 64, 0), (64, 64, 0), (0, 128, 0),
                   (128, 0, 44), (0, 0, 64), (64, 64, 44), (128, 64, 32)):
    print(f"  halved {nh:4d}  quartered {nq:4d}  in sixteenths {nf:4d}:  {mean_end(nh, nq, nf):6.1f} us")
