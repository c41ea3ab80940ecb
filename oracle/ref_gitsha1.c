/* Stand-in for the file the reference's CMake generates from src/utils/gitsha1.c.in
 * (CMakeLists.txt:69-71); /root/reference carries no git metadata, so the hash is "NoHash". */
char *gitHash(void) {
	return "NoHash";
}
