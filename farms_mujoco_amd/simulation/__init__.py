"""Simulation"""
