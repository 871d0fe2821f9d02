"""`from depthmotionnet.v2.networks import *` -- the reference's module path for the v2 model
(examples/example_v2.py:22), served by the MI355X-native path."""
