"""Step record — mirrors src/Gym/Observations/Step.cs:7-29 (Observation, Reward, Done, Information)."""
import numpy as np


class Step:
    __slots__ = ("Observation", "Reward", "Done", "Information")

    def __init__(self, observation=None, reward=0.0, done=False, information=None):   # Step.cs:15-20
        self.Observation = observation
        self.Reward = float(reward)
        self.Done = bool(done)
        self.Information = information

    def __iter__(self):            # Deconstruct, Step.cs:24-29: var (observation, reward, done, information) = ...
        yield self.Observation
        yield self.Reward
        yield self.Done
        yield self.Information

    def __eq__(self, other):       # Step.cs:35-39
        if not isinstance(other, Step):
            return NotImplemented
        return (np.array_equal(self.Observation, other.Observation) and self.Reward == other.Reward
                and self.Done == other.Done and self.Information == other.Information)

    def Clone(self):               # Step.cs:66-68
        return Step(np.array(self.Observation, copy=True), self.Reward, self.Done, self.Information)

    def __repr__(self):            # Step.cs:86-88
        return f"Reward: {self.Reward}, Done: {self.Done}, Information: {self.Information}, Observation: {self.Observation}"
