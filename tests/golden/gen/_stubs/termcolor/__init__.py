"""Stand-in for the third-party `termcolor` package (absent from this image).

Only used by tests/golden/gen/make_golden.py while importing the reference in the
build container; it is our own code, not reference code, and never ships to the GPU path.
"""


def colored(text, *args, **kwargs):
    return text
