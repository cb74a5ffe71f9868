"""Swimming"""
