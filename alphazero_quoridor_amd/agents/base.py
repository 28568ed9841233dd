"""agents/base.py of the reference: a named agent bound to a game; the default policy plays a
random legal pawn move."""
import numpy as np


class BaseAgent:
    def __init__(self, name, environment=None):
        self.name = name
        self.environment = environment

    def choose_action(self, game=None):
        game = game if game is not None else self.environment
        pawn = [a for a in game.actions() if a < 12]
        return int(np.random.choice(pawn))
