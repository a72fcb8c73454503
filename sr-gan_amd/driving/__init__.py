"""Steering-angle (driving) application: mirror of the reference's ``driving`` package."""
