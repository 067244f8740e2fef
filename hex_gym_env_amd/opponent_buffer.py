"""OpponentBuffer: the self-play opponent pool, shared by the single-env SelfPlayEnv mirror and
the vectorised VecSelfPlayEnv.

Semantics of the reference (minihex/SelfplayWrapper.py:39-144):
  * a fixed-size pool of opponent policies with one score each, plus a separate "best" policy
    (the one with the highest score ever given to `replace`);
  * per episode the opponent is the best policy with probability 0.8, otherwise a uniformly
    drawn pool member (:97-104);
  * in evaluation mode the pool is walked in order, one member per episode (:92-96); the
    reference stops advancing at the end of the pool (the last member is reused);
  * `replace(i, policy, score)` overwrites member i and promotes it to best when its score beats
    the best score so far (:125-136).
"""
import numpy as np


class OpponentBuffer:
    def __init__(self, members, scores, best, prob_best=0.8, verbose=False):
        self.members = list(members)
        self.scores = np.asarray(scores, dtype=float)
        if len(self.members) != len(self.scores):
            raise ValueError("one score per pool member")
        self.best = best
        self.best_score = float(np.max(self.scores)) if len(self.scores) else -np.inf
        self.best_mean_reward = -np.inf
        self.prob_best = prob_best
        self.evaluating = False
        self.eval_cursor = 0
        self.verbose = verbose
        # bumped by everything that changes WHICH policies play (replace / append / set_eval / assigning best or members):
        # what a captured collection step or a packed weights table was built from is named by this number, never by
        # ids of policy objects that may have been freed and their addresses re-used since
        self.generation = 0

    def __setattr__(self, name, value):
        if name in ("members", "best", "prob_best"):
            object.__setattr__(self, "generation", getattr(self, "generation", 0) + 1)
        object.__setattr__(self, name, value)

    def __len__(self):
        return len(self.members)

    # -- episode-level choice ------------------------------------------------------
    def pick(self, u_best, u_member):
        """slot for ONE new episode outside evaluation: -1 = the best policy, k = pool member k.
        u_best, u_member: two uniform [0,1) numbers (the caller owns the random stream)."""
        return -1 if u_best < self.prob_best else int(u_member * len(self.members))

    def next_eval_slot(self, wrap=False):
        """slot of the next evaluation episode, or None when a non-wrapping walk is exhausted"""
        if self.eval_cursor > len(self.members) - 1:
            if not wrap:
                return None
            self.eval_cursor = 0
        k = self.eval_cursor
        self.eval_cursor += 1
        return k

    def policy(self, slot):
        return self.best if slot < 0 else self.members[slot]

    # -- maintenance ----------------------------------------------------------------
    def set_eval(self, flag):
        self.eval_cursor = 0
        self.evaluating = bool(flag)
        self.generation += 1
        assert len(self.members) == len(self.scores)

    def replace(self, index, policy, score):
        if self.verbose:
            print("Previous model scores: ", self.scores)
        self.members[index] = policy
        self.scores[index] = score
        self.generation += 1
        if self.verbose:
            print("Replaced model from index: ", index)
            print("Model score: ", score)
        if score > self.best_score:
            self.best, self.best_score = policy, score
            if self.verbose:
                print("NEW BEST MODEL WITH SCORE: ", score)

    def append(self, policy, as_best=False, mean_reward=None):
        if as_best:
            self.best, self.best_mean_reward = policy, mean_reward
        self.members.append(policy)
        self.scores = np.append(self.scores, 0.0)
        self.generation += 1

    def save_best(self, path=None):
        self.best.save_model(path or ("models/best_model_" + str(self.best_score)))
