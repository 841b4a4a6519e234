"""Test environments for the worker protocol of VectorEnv: one that prints on stdout like an emulator stack does (banners at
import, warnings while stepping), one whose step never returns."""
import os
import sys
import time

from slimdqn.environments.synthetic import SyntheticAtariEnv

print("A.L.E: Arcade Learning Environment (a banner on stdout at import)")
os.write(1, b"Aa raw write to file descriptor 1\n")


class ChattyEnv(SyntheticAtariEnv):
    def step(self, action):
        print("S warning from the emulator: step", flush=True)
        sys.stdout.write("R")
        sys.stdout.flush()
        return super().step(action)


class HungEnv(SyntheticAtariEnv):
    def step(self, action):
        time.sleep(3600)
