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
