"""CPU: the direct path's scratch-slot table (zephyr_amd/csrc/capi.hip) is kept PER DEVICE -- the in-process counterpart of the reference's pool,
where every worker has an address space of its own (zephyr/backend/distributors.py:80-96,161-168).  helm_debug_ws_selftest drives the table's
own booking / lease code with host memory and logical devices, so the eight-GPU case is testable without a GPU."""
import pytest


@pytest.mark.parametrize('ndev,concurrent', [(1, 1), (2, 3), (8, 3), (8, 1), (16, 2)])
def test_every_logical_device_gets_its_own_booked_slots(helm_lib, ndev, concurrent):
    assert helm_lib.helm_debug_ws_selftest(ndev, concurrent, 1 << 16) == 0


def test_slot_count_follows_the_environment(helm_lib, monkeypatch):
    monkeypatch.setenv('HELM_WS_SLOTS', '4')
    assert helm_lib.helm_debug_ws_slots(-1, 0) == 4
    assert helm_lib.helm_debug_ws_selftest(8, 4, 4096) == 0
    monkeypatch.setenv('HELM_WS_SLOTS', '2')
    assert helm_lib.helm_debug_ws_slots(-1, 0) == 2
    assert helm_lib.helm_debug_ws_selftest(8, 3, 4096) == 0          # asks for more than there are: every device gets its two
    monkeypatch.setenv('HELM_WS_SLOTS', '99')
    assert helm_lib.helm_debug_ws_slots(-1, 0) == 4                  # clamped


def test_bad_arguments(helm_lib):
    assert helm_lib.helm_debug_ws_selftest(0, 1, 64) < 0
    assert helm_lib.helm_debug_ws_selftest(1, 0, 64) < 0
    assert helm_lib.helm_debug_ws_selftest(1, 1, 0) < 0
