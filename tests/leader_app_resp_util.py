"""Replays raft_test.go:2591-2657 TestLeaderAppResp (tests/golden/tables.json, lifted mechanically) through a
figure7-style driver: stepLeader's MsgAppResp branch end to end — Progress bookkeeping (raft.go:1362-1500),
raft.maybeCommit (:1502), bcastAppend after a commit advance (:1503-1504), sendAppend after a rejection or an
un-pausing ack (:1460,:1505-1509) and the `for r.maybeSendAppend(m.From, false)` loop (:1511-1520) — and checks what
the reference checks: Progress[2].Match / Next afterwards, the number of messages the leader sent, and every
message's Index (= prev index) and Commit.

The engine leaves composing and sending to the host: here the "host" is this function, which sends what the action
byte, the commit pass and the append plan ask for, and reports each send back as a SENT_APP record — the contract of
INTEGRATION.md §2a."""
import golden_util as gu
import progress_util as pu

po = pu.po


def load():
    t = gu.load_tables()["TestLeaderAppResp"]
    assert len(t["rows"]) == 4
    return t


def fixture_for_driver(t):
    """What figure7's drivers read: the voters and the log the leader starts from."""
    return {"voters": t["peers"], "log": t["log"]}


def replay_row(t, row, drv):
    me, frm = t["peers"][0], t["from"]
    last = t["log"][-1][0]
    term = 1  # newTestRaft starts at term 0; becomeCandidate makes it 1 (raft_test.go:2617-2625)
    # becomeLeader (raft.go:933-972): Progress reset with Next = lastIndex + 1, then the empty entry of the new term
    drv.become_leader(me, last, t["log"][0][0], 0, term)
    assert drv.step("appended", me, index=last + 1) == 0
    assert drv.step("app_resp", me, index=last + 1) & po.ACT_UPDATED
    last += 1
    committed = drv.maybe_commit()
    assert committed == 0
    st = drv.progress(frm)
    assert (st["match"], st["next"]) == (0, last), st  # "initial progress: match = 0; next = 3"

    sent = []  # (to, prev index, commit)

    def send(to, if_empty):
        plan = drv.plan(to)
        if plan == po.SEND_NONE or (plan == po.SEND_EMPTY and not if_empty):
            return False
        assert plan in (po.SEND_APPEND, po.SEND_EMPTY), plan
        nxt = drv.progress(to)["next"]
        n = last - nxt + 1 if plan == po.SEND_APPEND else 0  # everything from Next on fits one message here
        sent.append((to, nxt - 1, committed))
        drv.step("sent_app", to, entries=n, index=nxt - 1 + n, nbytes=n)
        return True

    act = drv.step("app_resp", frm, index=row["index"], reject=row["reject"], hint=row["index"], log_term=0,
                   resolved=row["index"])
    assert drv.progress(frm)["match"] == row["wmatch"], row
    if row["reject"]:
        if act & po.ACT_SEND_APPEND:  # raft.go:1460: MaybeDecrTo succeeded -> sendAppend
            send(frm, True)
    else:
        c = drv.maybe_commit()
        if c > committed:  # raft.go:1502-1504
            committed = c
            for to in t["peers"]:
                if to != me:
                    send(to, True)
        elif (act & (po.ACT_UPDATED | po.ACT_WAS_PAUSED)) == (po.ACT_UPDATED | po.ACT_WAS_PAUSED):  # :1505-1509
            send(frm, True)
        if act & po.ACT_UPDATED:
            while send(frm, False):  # :1511-1520
                pass
    assert drv.maybe_commit() == committed
    st = drv.progress(frm)
    assert st["next"] == row["wnext"], (row, st)
    assert len(sent) == row["wmsgNum"], (row, sent)
    for to, prev, commit in sent:
        assert prev == row["windex"] and commit == row["wcommitted"], (row, sent)
    return sent


def load_increase_next():
    t = gu.load_tables()["TestLeaderIncreaseNext"]
    assert len(t["rows"]) == 2
    return t


def replay_increase_next(t, row, drv):
    """raft_test.go:2760-2790 TestLeaderIncreaseNext: after one proposal the leader's bcastAppend leaves
    Progress[2].Next at last + 1 in StateReplicate (optimistic) and where it was in StateProbe."""
    me, peer = t["peers"]
    last, term = t["log"][-1][0], 1
    drv.become_leader(me, last, t["log"][0][0], 0, term)
    for _ in range(2):  # becomeLeader's empty entry, then the proposal (raft.go:787-819 appendEntry)
        last += 1
        assert drv.step("appended", me, index=last) == 0
        assert drv.step("app_resp", me, index=last) & po.ACT_UPDATED
        if last == t["log"][-1][0] + 1:
            drv.set_progress(peer, row["state"], row["next"])
    assert drv.maybe_commit() == 0  # two peers, one has nothing: no quorum
    plan = drv.plan(peer)  # bcastAppend -> maybeSendAppend(peer, true)
    assert plan == po.SEND_APPEND
    nxt = drv.progress(peer)["next"]
    assert nxt == row["next"]
    n = last - nxt + 1
    drv.step("sent_app", peer, entries=n, index=nxt - 1 + n, nbytes=n)
    assert drv.progress(peer)["next"] == row["wnext"], (row, drv.progress(peer))


def load_current_term_commit():
    t = gu.load_tables()["TestLeaderOnlyCommitsLogFromCurrentTerm"]
    assert len(t["rows"]) == 3
    return t


def replay_current_term_commit(t, row, drv):
    """raft_paper_test.go:871-901 TestLeaderOnlyCommitsLogFromCurrentTerm: a quorum index whose entry carries an
    earlier term does not move raftLog.committed (log.go:447-456); one of the leader's own term does."""
    me, peer = t["peers"]
    last = t["log"][-1][0]
    term = t["term_before_campaign"] + 1  # becomeCandidate (raft.go:887-899)
    drv.become_leader(me, last, t["log"][0][0], 0, term)
    for _ in range(2):  # becomeLeader's empty entry, then the proposal
        last += 1
        assert drv.step("appended", me, index=last) == 0
        assert drv.step("app_resp", me, index=last) & po.ACT_UPDATED
    assert drv.maybe_commit() == 0
    drv.step("app_resp", peer, index=row["index"])
    assert drv.maybe_commit() == row["wcommit"], row
