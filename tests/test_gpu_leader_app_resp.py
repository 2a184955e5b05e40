"""raft_test.go:2591-2657 TestLeaderAppResp on the engine through the C ABI: the response arrives as serialized
bytes (rafthip_step_wire), the commit pass is rafthip_commit_advance, the sends follow rafthip_plan_appends and
come back as SENT_APP records; several copies of the group with the voters in different slot orders must agree."""
import pytest

import leader_app_resp_util as lar
import raft_amd
from test_gpu_figure7 import EngineDriver

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("form", [raft_amd.capi.FLAG_STEP_LINKED, raft_amd.capi.FLAG_STEP_PARTITION], ids=["linked", "partitioned"])
@pytest.mark.parametrize("G", [1, 301])
def test_leader_app_resp_table_on_the_engine(G, form):
    t = lar.load()
    fx = lar.fixture_for_driver(t)
    for row in t["rows"]:
        with raft_amd.Engine(G, len(fx["voters"]), flags=form) as e:
            lar.replay_row(t, row, EngineDriver(e, fx, G, 256))


@pytest.mark.parametrize("form", [raft_amd.capi.FLAG_STEP_LINKED, raft_amd.capi.FLAG_STEP_PARTITION], ids=["linked", "partitioned"])
@pytest.mark.parametrize("G", [1, 130])
def test_leader_increase_next_table_on_the_engine(G, form):
    t = lar.load_increase_next()
    fx = lar.fixture_for_driver(t)
    for row in t["rows"]:
        with raft_amd.Engine(G, len(fx["voters"]), flags=form) as e:
            lar.replay_increase_next(t, row, EngineDriver(e, fx, G, 256))


@pytest.mark.parametrize("G", [1, 200])
def test_leader_only_commits_log_from_current_term_on_the_engine(G):
    """The term guard of rafthip_commit_advance against raft_paper_test.go:871-901."""
    t = lar.load_current_term_commit()
    fx = lar.fixture_for_driver(t)
    for row in t["rows"]:
        with raft_amd.Engine(G, len(fx["voters"])) as e:
            lar.replay_current_term_commit(t, row, EngineDriver(e, fx, G, 256))
