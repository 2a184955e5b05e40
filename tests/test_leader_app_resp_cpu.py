"""raft_test.go:2591-2657 TestLeaderAppResp on the oracle (see tests/leader_app_resp_util.py)."""
import leader_app_resp_util as lar
from test_figure7_cpu import OracleDriver


def test_leader_app_resp_table_on_the_oracle(oracle):
    t = lar.load()
    for row in t["rows"]:
        lar.replay_row(t, row, OracleDriver(oracle, lar.fixture_for_driver(t)))


def test_leader_increase_next_table_on_the_oracle(oracle):
    t = lar.load_increase_next()
    for row in t["rows"]:
        lar.replay_increase_next(t, row, OracleDriver(oracle, lar.fixture_for_driver(t)))


def test_leader_only_commits_log_from_current_term_on_the_oracle(oracle):
    t = lar.load_current_term_commit()
    for row in t["rows"]:
        lar.replay_current_term_commit(t, row, OracleDriver(oracle, lar.fixture_for_driver(t)))


def test_raftlog_term_table():
    """log_test.go:830-858 TestTerm pins the oracle's raftLog.term (log.go:381-407), the lookup under
    raftLog.maybeCommit's guard and findConflictByTerm."""
    import golden_util as gu
    from oracle import pyoracle as po
    t = gu.load_tables()["TestTerm"]
    assert len(t["rows"]) == 5
    offset, num = t["offset"], t["num"]
    terms = list(range(1, num))  # entries offset+i carry term i, i in [1, num) (log_test.go:837-839)
    err = {None: 0, "ErrCompacted": 1, "ErrUnavailable": 2}
    for row in t["rows"]:
        assert po.raftlog_term(offset + 1, 1, terms, row["idx"]) == (row["term"], err[row["err"]]), row
