"""Crowd counting application: mirror of the reference's ``crowd`` package."""
