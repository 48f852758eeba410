"""MI355X-native ORB front-end + Hamming matcher behind ORB-SLAM3's ORBextractor / ORBmatcher seam."""
__version__ = "0.1.0"
