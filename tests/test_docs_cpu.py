"""DESIGN.md states the current design and stays readable: at most 400 lines of at most 120 columns (the measured-and-
rejected experiments live in LABNOTES.md), and every file it names exists."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_design_is_short_and_narrow():
    lines = (ROOT / "DESIGN.md").read_text().splitlines()
    assert len(lines) <= 400
    wide = [(i + 1, len(l)) for i, l in enumerate(lines) if len(l) > 120]
    assert not wide, wide
    assert (ROOT / "LABNOTES.md").exists()


def test_design_names_files_that_exist():
    text = (ROOT / "DESIGN.md").read_text()
    for rel in set(re.findall(r"`((?:bgsa_amd|scripts|tests|oracle|include|examples)/[A-Za-z0-9_./]+\.(?:py|hip|h|c|inl|inc|sh|md))`", text)):
        assert (ROOT / rel).exists(), rel
