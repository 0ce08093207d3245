"""The lane compressor's workspace search (snappier_amd/csrc/piece_search.h -- the very code capi_pool.hip runs at the first large compress
call) against a model of device memory on the CPU: tests/abi/piece_search_model.cpp gives every candidate piece a share of each of three
kinds of memory and prices a probe as the microbenchmark measured it.  What must hold: the search finds a set spread
over three kinds when they turn up early, over two otherwise, keeps allocating while only one kind has been seen, stops at its
candidate limit, survives running out of memory, never picks a piece twice, and is never worse than taking the first sixteen.
(The first version of this model is what showed that the original stop rule -- largest share <= 0.52 / 0.36 -- was never met.)"""
import json
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def scenarios(tmp_path_factory):
    exe = tmp_path_factory.mktemp("piece_search") / "model"
    src = os.path.join(ROOT, "tests", "abi", "piece_search_model.cpp")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", src, "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    rows = [json.loads(line) for line in out.splitlines() if line.startswith("{")]
    return {r["scenario"]: r for r in rows}


def test_every_scenario_ran_and_picks_sixteen_distinct_candidates(scenarios):
    assert len(scenarios) == 11
    for name, r in scenarios.items():
        assert r["distinct"] and r["in_range"], name
        if name != "out of memory at 10":
            assert r["chosen"] == 16 and r["ms"] >= 0, name
            assert r["largest_share"] <= r["first_16_largest_share"] + 0.04, f"{name}: worse than the first sixteen pieces"


def test_three_kinds_are_used_when_they_turn_up_early(scenarios):
    r = scenarios["three kinds early"]
    assert r["references"] == 3 and r["largest_share"] <= 0.40 and r["candidates"] <= 48


def test_two_kinds_stop_the_search_after_six_rounds(scenarios):
    r = scenarios["two kinds early"]
    assert r["largest_share"] <= 0.56 and r["candidates"] == 96


def test_a_thorough_search_goes_on_to_the_third_kind(scenarios):
    r = scenarios["two kinds early, thorough"]
    assert r["references"] == 3 and r["largest_share"] <= 0.45 and 144 <= r["candidates"] <= 192


def test_weaker_contrast_between_the_levels_still_ends_balanced(scenarios):
    r = scenarios["two kinds early, weak contrast"]
    assert r["largest_share"] <= 0.56 and r["references"] >= 2            # levels 10 % apart instead of 18 %: still a balanced set


def test_the_search_goes_on_while_it_has_seen_one_kind_only(scenarios):
    r = scenarios["second kind after 110 candidates"]
    assert 112 <= r["candidates"] <= 144 and r["largest_share"] <= 0.56
    r = scenarios["one kind only"]
    assert r["candidates"] == 64 and r["largest_share"] == 1.0            # its limit; nothing better exists


def test_mixed_pieces_do_not_become_references(scenarios):
    """Round 4 on the GPU: a first round of pieces that lie across two kinds made four "references" of them and ended the search at once with a
    workspace at the two-kind level.  Only one-sided pieces may be references: the search goes on until the pure runs show up and ends on three kinds."""
    r = scenarios["mixed pieces first"]
    assert r["references"] == 3 and r["largest_share"] <= 0.45 and r["candidates"] >= 96


def test_pieces_that_are_balanced_by_themselves_are_enough(scenarios):
    assert scenarios["all pieces balanced"]["largest_share"] <= 0.56


def test_no_room_and_out_of_memory(scenarios):
    r = scenarios["no spare candidates"]
    assert r["candidates"] == 16 and r["probes"] == 0 and r["ms"] == 0      # the workspace is what could be allocated: nothing to measure
    assert scenarios["out of memory at 10"]["ms"] < 0                          # fewer pieces than the workspace needs: the caller reports it
    r = scenarios["out of memory at 20"]
    assert r["candidates"] == 20 and r["largest_share"] <= 0.56              # ... but a short supply of spares is still searched
