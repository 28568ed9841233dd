"""agents/manual.py of the reference: a human at the terminal, a human behind a GUI that
pushes the chosen action in (`receive_action`), and a replay of a recorded move list."""
from .base import BaseAgent


class ManualCLIAgent(BaseAgent):
    def __init__(self, name="human", environment=None, read=input, write=print):
        super().__init__(name, environment)
        self._read, self._write = read, write

    def choose_action(self, game=None):
        game = game if game is not None else self.environment
        self._write("Current Board")
        game.print_board()
        valid = game.actions()
        self._write("Available Actions")
        self._write(valid)
        while True:
            raw = self._read("Choose Action: ")
            try:
                action = int(raw)
            except ValueError:
                action = -1
            if action in valid:
                return action
            self._write("Invalid Action: {action} - please select a valid action".format(action=raw))
            self._write(valid)


class ManualPygameAgent(BaseAgent):
    def __init__(self, name):
        super().__init__(name)
        self._action = None

    def receive_action(self, action):
        self._action = action

    def choose_action(self, game=None):
        return self._action


class HistoricalAgent(BaseAgent):
    """Replays `moveset` (the reference's HistoricalPygameAgent, as a plain iterator)."""

    def __init__(self, name, moveset):
        super().__init__(name)
        self._moves = iter(moveset)

    def choose_action(self, game=None):
        return int(next(self._moves))
