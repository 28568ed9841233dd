"""Agents of the interactive front end (reference: agents/base.py, agents/manual.py)."""
from .base import BaseAgent
from .manual import HistoricalAgent, ManualCLIAgent, ManualPygameAgent

__all__ = ["BaseAgent", "ManualCLIAgent", "ManualPygameAgent", "HistoricalAgent"]
