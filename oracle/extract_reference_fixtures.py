#!/usr/bin/env python3
"""Copy the head of the known-answer DATA files the reference's own utility
tests hold into tests/golden/ (authoring container only; needs /root/reference).

  (src/img2body/{pixels,Matout}.txt is NOT used: Matout.txt was generated for other camera
   intrinsics than the ones compiled into src/img2body/img2body.cpp:31-35, e.g. pixel
   (514,228) -> 0.2524 there vs 3.16 from the program's own constants, so it pins nothing.)
  src/euler2qbw/test.sh:6-7   euler.txt   -> Mquat.txt    (MATLAB-generated)
  src/rotate/test.sh:8-10     testing/acc -> testing/acc.rot  (rotate -e 0 0 p)

These are data (inputs and expected outputs), not source.  They pin
Quaternion::rotation (src/ARC/Quaternion.cpp:32-47) and the Euler->quaternion
convention through independent, reference-authored answers.
"""
import itertools
import sys

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
PAIRS = [
    ("/root/reference/src/euler2qbw/euler.txt", "tests/golden/euler2qbw_euler.txt"),
    ("/root/reference/src/euler2qbw/Mquat.txt", "tests/golden/euler2qbw_mquat.txt"),
    ("/root/reference/src/rotate/testing/acc", "tests/golden/rotate_acc.txt"),
    ("/root/reference/src/rotate/testing/acc.rot", "tests/golden/rotate_acc_rot.txt"),
]
for src, dst in PAIRS:
    with open(src) as a, open(dst, "w") as b:
        b.writelines(itertools.islice(a, N))
    print(dst)
