"""MI355X-native PWCLO-Net point-cloud operator path."""
