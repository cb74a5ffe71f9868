"""Sensors"""
