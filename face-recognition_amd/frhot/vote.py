"""Threshold + identity vote over the top-5 hits of one query.

Host-side mirror of the decision the reference's DB service takes after its
Qdrant search (qdrant_database_FE/app.py:42-43,340-384).  Input: hits in
descending score order as (score, payload) with payload['id'].  Output: the JSON
content of the 200 answer.

Rule, as the reference computes it:
  1. keep hits with score >= 0.54; none -> "Point not found";
  2. count kept hits per id, ids in first-appearance order;
  3. more than one id and every count equal -> the best kept hit wins;
  4. otherwise the first id with the highest count wins; its score is the
     maximum over ALL hits of that id (kept or not) and its payload the first
     hit's payload with that id.
"""
from __future__ import annotations

THRESHOLD_PASS = 0.54


def vote(hits):
    kept = [(s, p) for s, p in hits if s >= THRESHOLD_PASS]
    if not kept:
        return {"message": "Point not found", "data": []}
    order, counts = [], {}
    for _s, p in kept:
        pid = p['id']
        if pid not in counts:
            counts[pid] = 0
            order.append(pid)
        counts[pid] += 1
    if len(order) > 1 and len({counts[i] for i in order}) == 1:
        best = kept[0]
        for h in kept[1:]:
            if h[0] > best[0]:
                best = h
        return {"message": "Point found", "data": [(best[0], best[1])]}
    top = max(counts[i] for i in order)
    winner = next(i for i in order if counts[i] == top)
    same = [(s, p) for s, p in hits if p['id'] == winner]
    return {"message": "Point found", "data": [(max(s for s, _ in same), same[0][1])]}
