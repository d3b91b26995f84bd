"""Caller-side mirror: the refinement loop object of the reference's pipelines/optimizer.py on the device-resident BatchRefiner."""
